// chol16_probe.hip — standalone check + timing of the in-LDS tile Cholesky solve (sadvio_amd/csrc/chol16.h), the same
// code k_solve<0> runs: random SPD systems of every size class against a host double-precision Cholesky, phase
// timestamps, hipEvent time per launch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/probe/_build/chol16_probe scripts/probe/chol16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../sadvio_amd/csrc/chol16.h"   // the product header (the round-6 experiments live in chol16_rows.h / chol16_pair.h with their own records in profiles/)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace sadvio;

template <int GATHER>
__global__ __launch_bounds__(512) void k_probe(const double* img, int N, double* xout, long long* ts, int* ok, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = c16_blocks(N + 1);
    const int sz = c16_size(N);
    double* A = (double*)smem;
    double* pub = A + sz;
    double* yv = pub + C16_WORK;
    double* xs = yv + 16 * nb;
    bool good = true;
    for (int rep = 0; rep < reps; rep++) {
        for (int i = threadIdx.x; i < sz; i += blockDim.x) A[i] = img[i];
        __syncthreads();
        c16_symmetrize(A, nb);
        __syncthreads();
        if (threadIdx.x == 0 && rep == reps - 1) ts[30] = clock64();
        good = c16_solve<GATHER>(A, N, xs, pub, yv, rep == reps - 1 ? ts : nullptr, (rep == reps - 1 && reps == 2) ? ts + 32 : nullptr) && good;
        __syncthreads();
        if (threadIdx.x == 0 && rep == reps - 1) ts[31] = clock64();
    }
    for (int i = threadIdx.x; i < N; i += blockDim.x) xout[i] = xs[i];
    if (threadIdx.x == 0) *ok = good ? 1 : 0;
}

static bool host_solve(std::vector<double> S, std::vector<double> b, int N, std::vector<double>& x) {
    for (int j = 0; j < N; j++) {
        double d = S[j * N + j];
        for (int q = 0; q < j; q++) d -= S[j * N + q] * S[j * N + q];
        if (!(d > 0)) return false;
        d = sqrt(d); S[j * N + j] = d;
        for (int i = j + 1; i < N; i++) {
            double t = S[i * N + j];
            for (int q = 0; q < j; q++) t -= S[i * N + q] * S[j * N + q];
            S[i * N + j] = t / d;
        }
    }
    for (int i = 0; i < N; i++) { double t = b[i]; for (int q = 0; q < i; q++) t -= S[i * N + q] * b[q]; b[i] = t / S[i * N + i]; }
    for (int i = N - 1; i >= 0; i--) { double t = b[i]; for (int q = i + 1; q < N; q++) t -= S[q * N + i] * b[q]; b[i] = t / S[i * N + i]; }
    x = b;
    return true;
}

int main() {
    CK(hipSetDevice(0));
    const int sizes[] = {114, 120, 165, 174, 1, 3, 5, 15, 16, 17, 31, 32, 33, 48, 60, 96, 128, 160, 113, 112};
    srand(7);
    for (int N : sizes) {
        std::vector<double> S((size_t)N * N), b(N), x;
        // S = G G^T / N + diag: well conditioned but dense
        std::vector<double> G((size_t)N * N);
        for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
        for (int i = 0; i < N; i++)
            for (int j = 0; j <= i; j++) {
                double t = 0;
                for (int q = 0; q < N; q++) t += G[(size_t)i * N + q] * G[(size_t)j * N + q];
                S[(size_t)i * N + j] = S[(size_t)j * N + i] = t / N + (i == j ? 0.05 : 0.0);
            }
        for (auto& v : b) v = (double)rand() / RAND_MAX - 0.5;
        if (!host_solve(S, b, N, x)) { printf("N=%d host solve failed\n", N); continue; }
        const int sz = c16_size(N), nb = c16_blocks(N + 1);
        std::vector<double> img(sz, 0.0);
        for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) img[c16_index(i, j)] = S[(size_t)i * N + j];
        for (int j = 0; j < N; j++) img[c16_index(N, j)] = b[j];
        // stale values where the assembly never writes (upper halves of the diagonal tiles are overwritten by symmetrize;
        // padding rows / columns must be harmless whatever they hold as long as they are finite zeros: the library zeroes S)
        double *dimg, *dx; long long* dts; int* dok;
        CK(hipMalloc(&dimg, sz * 8)); CK(hipMalloc(&dx, (N + 1) * 8)); CK(hipMalloc(&dts, 128 * 8)); CK(hipMalloc(&dok, 4));
        CK(hipMemcpy(dimg, img.data(), sz * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dts, 0, 128 * 8));
        const size_t lds = (size_t)(sz + C16_WORK + 16 * nb + 16 * nb + 16) * 8;
      for (int var = 0; var < 2; var++) {
        auto kern = var ? k_probe<0> : k_probe<1>;   // 1: the product (pivot entries gathered through LDS), 2: GATHER = 0 (v_readlane + v_permlane swaps)
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(1), dim3(512), lds, 0, dimg, N, dx, dts, dok, 3);
        CK(hipDeviceSynchronize());
        std::vector<double> xg(N); long long ts[128]; int ok = 0;
        CK(hipMemcpy(xg.data(), dx, N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ts, dts, 128 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost));
        double err = 0, mx = 0;
        for (int i = 0; i < N; i++) { err = fmax(err, fabs(xg[i] - x[i])); mx = fmax(mx, fabs(x[i])); }
        // residual of the GPU solution
        double res = 0;
        for (int i = 0; i < N; i++) { double t = -b[i]; for (int j = 0; j < N; j++) t += S[(size_t)i * N + j] * xg[j]; res = fmax(res, fabs(t)); }
        long long ts2[128];
        if (N == 114 || N == 165 || N == 120 || N == 174) {   // second launch with the in-step stamps (they cost ~55 cycles each: not in the timed run)
            hipLaunchKernelGGL(kern, dim3(1), dim3(512), lds, 0, dimg, N, dx, dts, dok, 2);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ts2, dts, 128 * 8, hipMemcpyDeviceToHost));
        }
        printf("N=%3d variant=%d ok=%d max|x - x_host| %.2e (|x| %.2e) residual %.2e  solve %lld cyc = %.2f us", N, var + 1, ok, err, mx, res, ts[31] - ts[30], (ts[31] - ts[30]) / 2400.0);
        if (N == 114 || N == 165 || N == 120 || N == 174) {
            printf("\n      block phases (cycles: replay | look-ahead+trailing):");
            const int nbc = c16_blocks(N);
            for (int kb = 0; kb < nbc && kb < 8; kb++) printf(" %lld|%lld", ts[1 + 2 * kb] - (kb ? ts[2 * kb] : ts[0]), ts[2 + 2 * kb] - ts[1 + 2 * kb]);
            if (var == 0) {
                printf("\n      last pivot block, per step [pairs gather chol y mfma Mpad]:");
                for (int st = 0; st < 4; st++) { printf(" |"); for (int q = 1; q <= 6; q++) printf(" %lld", ts2[32 + 8 * st + q] - ts2[32 + 8 * st + q - 1]); }
            }
            printf("\n     ");
            printf("  first block %lld, back-substitution %lld", ts[0] - ts[30], ts[20] - ts[2 * (nbc < 8 ? nbc : 8)]);
        }
        printf("\n");
        if (N == 114) {   // launch-to-launch time incl. the LDS fill, 200 launches
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(kern, dim3(1), dim3(512), lds, 0, dimg, N, dx, dts, dok, 1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("      N=114: %.2f us per launch (fill + symmetrize + solve + write-back, 200 back-to-back launches)\n", 1e3 * ms / 200);
        }
      }
        CK(hipFree(dimg)); CK(hipFree(dx)); CK(hipFree(dts)); CK(hipFree(dok));
    }
    return 0;
}
