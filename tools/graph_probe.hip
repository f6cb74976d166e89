// Launch floor of a dependent chain of tiny kernels on one stream: plain launches against the same chain captured in a hipGraph.
//   hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o tools/graph_probe && tools/graph_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void empty_kernel() {}
__global__ void touch_kernel(double *p, int n) {           // every work-group dirties a few cache lines (a release has work to do)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.0;
}
// producer / consumer pair over `n` doubles: what a GEMM's partial sums and the kernel that adds them do to each other
__global__ void produce_kernel(double *p, long n, double v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v + i;
}
__global__ void consume_kernel(const double *p, long n, double *out) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
int main() {
    const int N = 2000;
    const long big = 2L << 20;        // 16 MB of doubles
    double *buf; CK(hipMalloc(&buf, sizeof(double) * (big + 16))); CK(hipMemset(buf, 0, sizeof(double) * (big + 16)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int variant = 0; variant < 5; ++variant) {
        auto chain = [&](hipStream_t st) {
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
                else if (variant == 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st);
                else if (variant == 2) hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), 0, st, buf, 1 << 16);
                else {
                    const long n = variant == 3 ? big : big / 8;
                    if (i & 1) hipLaunchKernelGGL(consume_kernel, dim3(256), dim3(256), 0, st, buf, n, buf + big);
                    else hipLaunchKernelGGL(produce_kernel, dim3(256), dim3(256), 0, st, buf, n, (double)i);
                }
            }
        };
        const char *names[5] = {"empty, 1 work-group", "empty, 256 work-groups of 256", "256 work-groups each writing 2 KB",
                                "write 16 MB / read 16 MB, alternating", "write 2 MB / read 2 MB, alternating"};
        chain(s); CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventRecord(a, s)); chain(s); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
        const double plain = 1e3 * ms / N;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(s); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, a, b));
        printf("%-36s plain launches %6.2f us per kernel | hipGraph %6.2f us per kernel\n", names[variant], plain, 1e3 * ms / N);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
