// Decodes the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 empirically: for every pair (la, lb) of lanes,
// A = indicator(lane == la), B = indicator(lane == lb), C = 0; the set of non-zero output lanes tells which
// (block, i, k) lane la of A and which (block, k, j) lane lb of B hold.  Also times a dependent / independent
// stream of the instruction against v_mfma_f64_16x16x4_f64.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma4x4_probe.hip -o tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned long long *mask) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) mask[la * 64 + lb] = m;
        }
}

template <int MODE> __global__ void rate(double *out, int iters) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    if (MODE == 0) {         // 16 independent accumulators of 4x4x4
        double c[16];
        for (int i = 0; i < 16; ++i) c[i] = 0.0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
        double s = 0; for (int i = 0; i < 16; ++i) s += c[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {                  // 4 independent accumulators of 16x16x4
        d4_t c[4];
        for (int i = 0; i < 4; ++i) c[i] = (d4_t){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        double s = 0; for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

int main() {
    unsigned long long *mask;
    hipMalloc(&mask, sizeof(unsigned long long) * 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mask);
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), mask, sizeof(unsigned long long) * 4096, hipMemcpyDeviceToHost);
    // for A lane la: the B lanes it meets and the output lanes
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            const unsigned long long m = h[la * 64 + lb];
            if (!m) continue;
            printf(" B%d->D", lb);
            for (int l = 0; l < 64; ++l) if (m >> l & 1) printf("%d,", l);
        }
        printf("\n");
    }
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double flops = mode == 0 ? 1024.0 * 4 * iters * 16 * 512 : 1024.0 * 4 * iters * 4 * 2048;
            printf("%s: %.3f ms, %.1f TFLOP/s\n", mode == 0 ? "4x4x4 (16 acc)" : "16x16x4 (4 acc)", ms, flops / ms * 1e-9);
        }
    return 0;
}
