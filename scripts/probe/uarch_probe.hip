// uarch_probe.hip — measured constants behind the design of the in-LDS reduced-system solver (k_solve, N_p <= 174):
// FP64 dependent-chain latencies, v_rsq_f64 accuracy, v_mfma_f64_16x16x4_f64 operand / result layout and latency,
// v_readlane / permlane-swap cost, s_barrier and LDS-flag hand-off cost between waves of one workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probe/_build/uarch_probe scripts/probe/uarch_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long clk() { return __builtin_readcyclecounter(); }
// s_memtime ordered after everything `x` depends on has ISSUED (the VALU pipe is in order) and before anything that uses x
__device__ __forceinline__ long long clk_after(double& x) {
    long long t;
    asm volatile("s_nop 0" : "+v"(x) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    asm volatile("s_nop 0" : "+v"(x) :: "memory");
    return t;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor16_other(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return (threadIdx.x & 16) ? __hiloint2double(b[0], a[0]) : __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xor32_other(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return (threadIdx.x & 32) ? __hiloint2double(b[0], a[0]) : __hiloint2double(b[1], a[1]);
}

// ---- 1. rsq accuracy ----
__global__ void k_rsq(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double y0 = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-d * y0, y0, 1.0);
    double y1 = __builtin_fma(0.5 * y0, e, y0);
    e = __builtin_fma(-d * y1, y1, 1.0);
    double y2 = __builtin_fma(0.5 * y1, e, y1);
    // Goldschmidt, 2 rounds: g -> sqrt, h -> 1 / (2 sqrt)
    double g = d * y0, h = 0.5 * y0;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
    out[4 * i] = y0; out[4 * i + 1] = y1; out[4 * i + 2] = y2; out[4 * i + 3] = 2.0 * h;
}

// ---- 2. MFMA layout: D = A(16x4) B(4x16), A[i][k] = 1 + i + 100 k, B[k][j] = (k == kk) * (j == jj) probes ----
__global__ void k_mfma_layout(double* out) {
    const int ln = threadIdx.x;
    // hypothesis: A operand lane l holds A[l % 16][l / 16]; B operand lane l holds B[l / 16][l % 16]
    const double a = 1.0 + (ln % 16) + 100.0 * (ln / 16);   // A[i][k] = 1 + i + 100 k
    const double b = 1.0 + 0.001 * (ln % 16) + 7.0 * (ln / 16);   // B[k][j] = 1 + 0.001 j + 7 k
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[4 * ln + r] = c[r];
}

// ---- 3. latencies (one wave unless stated) ----
template <int N>
__device__ __forceinline__ double chain_fma(double x, double a, double b) {
#pragma unroll
    for (int i = 0; i < N; i++) x = __builtin_fma(x, a, b);
    return x;
}

__global__ __launch_bounds__(512) void k_lat(double* out, long long* cyc, const double* in) {
    __shared__ double sh[4096];
    __shared__ volatile int flag[64];
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    double x = in[ln], a = in[64], b = in[65];
    long long t0, t1;
    if (tid < 64) flag[tid] = 0;
    __syncthreads();
    // (0) dependent FP64 FMA chain, 256 deep
    if (wv == 0) {
        t0 = clk_after(x); x = chain_fma<256>(x, a, b); t1 = clk_after(x);
        if (ln == 0) cyc[0] = t1 - t0;
        out[ln] = x;
    }
    __syncthreads();
    // (1) 4 independent FMA chains interleaved, 64 deep each (issue rate)
    if (wv == 0) {
        double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
        t0 = clk_after(y3);
#pragma unroll
        for (int i = 0; i < 64; i++) { y0 = __builtin_fma(y0, a, b); y1 = __builtin_fma(y1, a, b); y2 = __builtin_fma(y2, a, b); y3 = __builtin_fma(y3, a, b); }
        y3 += y0 + y1 + y2;
        t1 = clk_after(y3);
        if (ln == 0) cyc[1] = t1 - t0;
        out[64 + ln] = y0 + y1 + y2 + y3;
    }
    __syncthreads();
    // (2) dependent rsq chain (rsq -> fma -> rsq ...), 64 deep: cost of rsq + 1 fma
    if (wv == 0) {
        double y = fabs(x) + 1.0;
        t0 = clk_after(y);
#pragma unroll
        for (int i = 0; i < 64; i++) { y = __builtin_amdgcn_rsq(y); y = __builtin_fma(y, a, 2.0); }
        t1 = clk_after(y);
        if (ln == 0) cyc[2] = t1 - t0;
        out[128 + ln] = y;
    }
    __syncthreads();
    // (3) full pivot step chain: rsq + 2 Newton + scale + dependent update, 32 deep
    if (wv == 0) {
        double d = fabs(x) + 2.0, l = 0.3;
        t0 = clk_after(d);
#pragma unroll
        for (int i = 0; i < 32; i++) {
            double y = __builtin_amdgcn_rsq(d);
            double e = __builtin_fma(-d * y, y, 1.0);
            y = __builtin_fma(0.5 * y, e, y);
            e = __builtin_fma(-d * y, y, 1.0);
            y = __builtin_fma(0.5 * y, e, y);
            l = l * y;
            d = __builtin_fma(-l, l, 3.0);
        }
        t1 = clk_after(d);
        if (ln == 0) cyc[3] = t1 - t0;
        out[192 + ln] = d;
    }
    __syncthreads();
    // (4) readlane -> VALU use -> readlane chain, 64 deep (value moves lane to lane)
    if (wv == 0) {
        double y = x;
        t0 = clk_after(y);
#pragma unroll
        for (int i = 0; i < 64; i++) { double s = readlane_f64(y, (i * 7 + 3) & 63); y = __builtin_fma(y, a, s); }
        t1 = clk_after(y);
        if (ln == 0) cyc[4] = t1 - t0;
        out[256 + ln] = y;
    }
    __syncthreads();
    // (5) 20 independent readlanes (10 doubles) + one dependent fma each: issue cost
    if (wv == 0) {
        double acc = x;
        t0 = clk_after(acc);
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 10; i++) acc += readlane_f64(x, (i * 5 + r) & 63);
        }
        t1 = clk_after(acc);
        if (ln == 0) cyc[5] = t1 - t0;
        out[320 + ln] = acc;
    }
    __syncthreads();
    // (6) dependent MFMA chain on one accumulator, 32 deep
    if (wv == 0) {
        d4 c = {x, x, x, x};
        double c0s = c[0];
        t0 = clk_after(c0s); c[0] = c0s;
#pragma unroll
        for (int i = 0; i < 32; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c0s = c[0]; t1 = clk_after(c0s); c[0] = c0s;
        if (ln == 0) cyc[6] = t1 - t0;
        out[384 + ln] = c[0] + c[1] + c[2] + c[3];
    }
    __syncthreads();
    // (7) 4 independent accumulators x 8 (MFMA issue rate)
    if (wv == 0) {
        d4 c0 = {x, x, x, x}, c1 = c0, c2 = c0, c3 = c0;
        double q0 = c3[0];
        t0 = clk_after(q0); c3[0] = q0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        q0 = c0[0] + c1[0] + c2[0] + c3[0]; t1 = clk_after(q0); c3[0] = q0;
        if (ln == 0) cyc[7] = t1 - t0;
        out[448 + ln] = c0[0] + c1[1] + c2[2] + c3[3];
    }
    __syncthreads();
    // (8) MFMA whose A/B operand depends on the previous MFMA's result through one VALU op (the within-block step): 16 deep
    if (wv == 0) {
        d4 c = {x, x, x, x};
        double y = a;
        t0 = clk_after(y);
#pragma unroll
        for (int i = 0; i < 16; i++) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0); y = c[i & 3] * b; }
        t1 = clk_after(y);
        if (ln == 0) cyc[8] = t1 - t0;
        out[512 + ln] = y;
    }
    __syncthreads();
    // (9) xor16 + xor32 exchange chain (gather of the 4 lane-group values), 32 deep
    if (wv == 0) {
        double y = x;
        t0 = clk_after(y);
#pragma unroll
        for (int i = 0; i < 32; i++) { double p = xor16_other(y); double q = xor32_other(y + p); y = __builtin_fma(q, a, p); }
        t1 = clk_after(y);
        if (ln == 0) cyc[9] = t1 - t0;
        out[576 + ln] = y;
    }
    __syncthreads();
    // (10) s_barrier, 512 threads, 64 in a row
    t0 = clk();
#pragma unroll 1
    for (int i = 0; i < 64; i++) __syncthreads();
    t1 = clk();
    if (tid == 0) cyc[10] = t1 - t0;
    // (11) LDS write -> barrier -> LDS read of another wave's data -> barrier, 32 rounds
    t0 = clk();
    double v = x;
#pragma unroll 1
    for (int i = 0; i < 32; i++) {
        sh[tid] = v;
        __syncthreads();
        v = sh[(tid + 64) & 511] + 1.0;
        __syncthreads();
    }
    t1 = clk();
    if (tid == 0) cyc[11] = t1 - t0;
    out[640 + (tid & 63)] = v;
    // (12) LDS flag ping-pong between wave 0 and wave 1 (no barrier): 64 round trips
    t0 = clk();
    if (wv == 0) {
#pragma unroll 1
        for (int i = 1; i <= 64; i++) {
            if (ln == 0) { flag[0] = i; }
            while (flag[1] < i) { }
        }
    } else if (wv == 1) {
#pragma unroll 1
        for (int i = 1; i <= 64; i++) {
            while (flag[0] < i) { }
            if (ln == 0) { flag[1] = i; }
        }
    }
    t1 = clk();
    if (tid == 0) cyc[12] = t1 - t0;
    __syncthreads();
    // (13) dependent LDS read chain (pointer chase), 64 deep, one wave
    if (wv == 0) {
        for (int i = ln; i < 4096; i += 64) sh[i] = (double)((i * 37 + 11) & 4095);
        __builtin_amdgcn_s_waitcnt(0);
        int idx = ln;
        t0 = clk();
#pragma unroll 1
        for (int i = 0; i < 64; i++) idx = (int)sh[idx];
        t1 = clk();
        if (ln == 0) cyc[13] = t1 - t0;
        out[704 + ln] = idx;
    }
    __syncthreads();
    // (14) LDS tile round trip in one wave: 4 x ds_write_b64 + wait + 4 x ds_read_b64 (transposing a 16x16 tile), 16 rounds
    if (wv == 0) {
        double r0 = x, r1 = x + 1, r2 = x + 2, r3 = x + 3;
        const int lr = ln & 15, lk = ln >> 4;
        t0 = clk();
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            sh[(lk + 0) * 20 + lr * 1 + 0] = r0;  // dummy addresses, conflict pattern of a padded transpose
            sh[(lk + 4) * 20 + lr] = r1; sh[(lk + 8) * 20 + lr] = r2; sh[(lk + 12) * 20 + lr] = r3;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            r0 = sh[lr * 20 + lk]; r1 = sh[lr * 20 + lk + 4]; r2 = sh[lr * 20 + lk + 8]; r3 = sh[lr * 20 + lk + 12];
            r0 += 1.0;
        }
        t1 = clk();
        if (ln == 0) cyc[14] = t1 - t0;
        out[768 + ln] = r0 + r1 + r2 + r3;
    }
}

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("device: %s, clock %d kHz, CUs %d\n", prop.gcnArchName, prop.clockRate, prop.multiProcessorCount);
    // 1. rsq accuracy
    {
        const int n = 1 << 20;
        std::vector<double> x(n), o(4 * n);
        srand(1);
        for (int i = 0; i < n; i++) x[i] = exp(((double)rand() / RAND_MAX - 0.5) * 60.0) * (1.0 + (double)rand() / RAND_MAX);
        double *dx, *dout;
        CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dout, 4 * n * 8));
        CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
        CK(hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost));
        double worst[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; i++) {
            long double ref = 1.0L / sqrtl((long double)x[i]);
            for (int q = 0; q < 4; q++) { double e = (double)fabsl(((long double)o[4 * i + q] - ref) / ref); if (e > worst[q]) worst[q] = e; }
        }
        printf("rsq_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e, Goldschmidt-2 %.3e (eps = 1.1e-16)\n", worst[0], worst[1], worst[2], worst[3]);
    }
    // 2. MFMA layout
    {
        double* dout; CK(hipMalloc(&dout, 256 * 8));
        hipLaunchKernelGGL(k_mfma_layout, dim3(1), dim3(64), 0, 0, dout);
        std::vector<double> o(256);
        CK(hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost));
        // reference D[i][j] = sum_k A[i][k] B[k][j]
        int okA = 1, okB = 1;
        for (int ln = 0; ln < 64; ln++)
            for (int r = 0; r < 4; r++) {
                const int lr = ln % 16, lk = ln / 16;
                auto D = [&](int i, int j) { double s = 0; for (int k = 0; k < 4; k++) s += (1.0 + i + 100.0 * k) * (1.0 + 0.001 * j + 7.0 * k); return s; };
                if (fabs(o[4 * ln + r] - D(lk + 4 * r, lr)) > 1e-9) okA = 0;   // row = lk + 4 r, col = lr
                if (fabs(o[4 * ln + r] - D(4 * lk + r, lr)) > 1e-9) okB = 0;   // row = 4 lk + r, col = lr
            }
        printf("mfma_f64_16x16x4 D layout: row = lane/16 + 4*reg : %s ; row = 4*(lane/16) + reg : %s (col = lane%%16)\n", okA ? "YES" : "no", okB ? "YES" : "no");
    }
    // 3. latencies
    {
        double *dout, *din; long long* dc;
        CK(hipMalloc(&dout, 2048 * 8)); CK(hipMalloc(&din, 128 * 8)); CK(hipMalloc(&dc, 32 * 8));
        std::vector<double> in(128);
        for (int i = 0; i < 128; i++) in[i] = 0.5 + 0.001 * i;
        in[64] = 0.999; in[65] = 0.001;
        CK(hipMemcpy(din, in.data(), 128 * 8, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_lat, dim3(1), dim3(512), 0, 0, dout, dc, din);
        CK(hipDeviceSynchronize());
        long long c[32];
        CK(hipMemcpy(c, dc, 32 * 8, hipMemcpyDeviceToHost));
        printf("cycles (s_memtime): \n");
        printf("  dependent v_fma_f64 chain            : %.1f / op\n", c[0] / 256.0);
        printf("  4 independent v_fma_f64 chains       : %.1f / op (issue)\n", c[1] / 256.0);
        printf("  rsq + fma dependent                  : %.1f / pair\n", c[2] / 64.0);
        printf("  pivot column chain (rsq+2NR+scale+upd): %.1f / column\n", c[3] / 32.0);
        printf("  readlane_f64 -> fma -> readlane chain: %.1f / step\n", c[4] / 64.0);
        printf("  10 x readlane_f64 + add (independent): %.1f / group of 10\n", c[5] / 8.0);
        printf("  dependent mfma_f64_16x16x4 chain     : %.1f / mfma\n", c[6] / 32.0);
        printf("  4 independent mfma accumulators      : %.1f / mfma (issue)\n", c[7] / 32.0);
        printf("  mfma -> valu -> mfma (operand dep)   : %.1f / round\n", c[8] / 16.0);
        printf("  xor16 + xor32 exchange + fma chain   : %.1f / round\n", c[9] / 32.0);
        printf("  s_barrier (512 threads)              : %.1f / barrier\n", c[10] / 64.0);
        printf("  lds write, barrier, read, barrier    : %.1f / round\n", c[11] / 32.0);
        printf("  LDS flag ping-pong wave0 <-> wave1   : %.1f / round trip\n", c[12] / 64.0);
        printf("  dependent ds_read chain              : %.1f / load\n", c[13] / 64.0);
        printf("  16x16 tile LDS transpose round trip  : %.1f / round\n", c[14] / 16.0);
    }
    return 0;
}
