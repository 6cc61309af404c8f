// probe: rocSOLVER dpotrf/dpotrs availability + timing on gfx950 (not part of the product)
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <vector>
#include <cstdio>
#include <cmath>
#include <chrono>
int main() {
    rocblas_handle hb; 
    auto t0 = std::chrono::steady_clock::now();
    if (rocblas_create_handle(&hb) != rocblas_status_success) { printf("no handle\n"); return 1; }
    printf("create_handle %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    hipStream_t st; hipStreamCreate(&st); rocblas_set_stream(hb, st);
    for (int N : {300, 600, 1065, 1500, 3000}) {
        std::vector<double> A((size_t)N * N), b(N, 1.0);
        for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) A[i + (size_t)j * N] = (i == j ? N : 0.0) + 1.0 / (1 + abs(i - j));
        double *dA, *dA0, *dB; int* info;
        hipMalloc(&dA, sizeof(double) * N * N); hipMalloc(&dA0, sizeof(double) * N * N); hipMalloc(&dB, sizeof(double) * N); hipMalloc(&info, 4);
        hipMemcpy(dA0, A.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
        hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        float best_f = 1e9, best_s = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipMemcpyAsync(dA, dA0, sizeof(double) * N * N, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(dB, b.data(), sizeof(double) * N, hipMemcpyHostToDevice, st);
            hipEventRecord(e0, st);
            rocsolver_dpotrf(hb, rocblas_fill_lower, N, dA, N, info);
            hipEventRecord(e1, st);
            rocsolver_dpotrs(hb, rocblas_fill_lower, N, 1, dA, N, dB, N);
            hipEventRecord(e2, st);
            hipStreamSynchronize(st);
            float f, s; hipEventElapsedTime(&f, e0, e1); hipEventElapsedTime(&s, e1, e2);
            if (rep) { best_f = fmin(best_f, f); best_s = fmin(best_s, s); }
        }
        std::vector<double> x(N); hipMemcpy(x.data(), dB, sizeof(double) * N, hipMemcpyDeviceToHost);
        double res = 0; for (int i = 0; i < N; i++) { double r = -1; for (int j = 0; j < N; j++) r += A[i + (size_t)j * N] * x[j]; res = fmax(res, fabs(r)); }
        int hi; hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost);
        printf("N=%d potrf %.3f ms (%.2f TFLOP/s) potrs %.3f ms info %d resid %.2e\n", N, best_f, N * 1.0 * N * N / 3 / best_f / 1e9, best_s, hi, res);
        hipFree(dA); hipFree(dA0); hipFree(dB); hipFree(info);
    }
    return 0;
}
