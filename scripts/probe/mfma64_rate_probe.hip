// Is the issue rate of v_mfma_f64_16x16x4_f64 a per-SIMD or a per-CU resource on gfx950? One workgroup of 1 .. 8 waves, every wave
// runs N MFMAs on four independent accumulators; cycles per MFMA per wave, and aggregate MFMAs per cycle for the CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probe/_build/mfma64_rate_probe scripts/probe/mfma64_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* cyc, int n) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; i += 4) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const long long t1 = clock64();
    __syncthreads();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
__global__ void kv(double* out, long long* cyc, int n) {   // the same amount of arithmetic as v_fma_f64: 16 per MFMA and lane
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int u = 0; u < 16; u++) a[u & 7] = __builtin_fma(x, y, a[u & 7]);
    const long long t1 = clock64();
    __syncthreads();
    out[threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h[16];
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 8);
    const int n = 4000;
    for (int waves = 1; waves <= 16; waves *= 2) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
        printf("mfma_f64_16x16x4: %2d wave(s) x %d MFMAs: %lld cycles -> %.1f cycles / MFMA / wave, %.3f MFMA / cycle / CU = %.0f flop / cycle / CU\n",
               waves, n, mx, (double)mx / n, (double)waves * n / mx, 2048.0 * waves * n / mx);
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(kv, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        mx = 0; for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
        printf("v_fma_f64       : %2d wave(s) x %d x 16 FMAs: %lld cycles -> %.0f flop / cycle / CU\n", waves, n, mx, 2048.0 * waves * n / mx);
    }
    return 0;
}
