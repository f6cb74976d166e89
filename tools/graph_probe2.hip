// A step-sized graph (6 kernel nodes in a chain) replayed with fresh kernel arguments every time: host cost of
// hipGraphExecKernelNodeSetParams x 6 + hipGraphLaunch, and the device time per kernel, against six plain launches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work_kernel(double *p, long n, double v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = p[i] * 0.5 + v;
}
int main() {
    const int NODES = 6, ITER = 2000;
    const long n = 1L << 18;     // 2 MB
    double *buf; CK(hipMalloc(&buf, sizeof(double) * n)); CK(hipMemset(buf, 0, sizeof(double) * n));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    // plain
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(a, s));
        for (int it = 0; it < ITER; ++it)
            for (int k = 0; k < NODES; ++k) hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, s, buf, n, (double)(it + k));
        CK(hipEventRecord(b, s));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("plain launches:          device %6.2f us per kernel, host enqueue %6.2f us per step of %d kernels\n", 1e3 * ms / (ITER * NODES),
                        std::chrono::duration<double, std::micro>(t1 - t0).count() / ITER, NODES);
    }
    // explicit graph, params updated per replay
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes(NODES);
    double *pb = buf; long nn = n; double v = 0.0;
    void *args[3] = {&pb, &nn, &v};
    hipKernelNodeParams kp = {};
    kp.func = (void *)work_kernel; kp.gridDim = dim3(256); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
    for (int k = 0; k < NODES; ++k) CK(hipGraphAddKernelNode(&nodes[k], g, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &kp));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(a, s));
        for (int it = 0; it < ITER; ++it) {
            for (int k = 0; k < NODES; ++k) { v = (double)(it + k); CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp)); }
            CK(hipGraphLaunch(ge, s));
        }
        CK(hipEventRecord(b, s));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("graph, params set/replay: device %6.2f us per kernel, host enqueue %6.2f us per step of %d kernels\n", 1e3 * ms / (ITER * NODES),
                        std::chrono::duration<double, std::micro>(t1 - t0).count() / ITER, NODES);
    }
    double chk; CK(hipMemcpy(&chk, buf, 8, hipMemcpyDeviceToHost)); printf("check %.6f\n", chk);
    return 0;
}
