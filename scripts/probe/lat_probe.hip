// lat_probe.hip — issue / dependent-latency figures of the instructions the row-layout pivot block of chol16.h is made of
// (round 6): FP64 fma, v_rsq_f64, v_fmac_f64_dpp row_newbcast, v_readlane -> SGPR -> VALU, v_permlane16_swap. One wave, volatile
// asm chains, s_memtime around 1 024 instances. Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probe/_build/lat_probe scripts/probe/lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

__device__ __forceinline__ long long now() {
    long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

#define TEST(ID, N, DECL, BODY, SINK)                                        \
    {                                                                         \
        DECL;                                                                 \
        long long t0 = now();                                                 \
        for (int it = 0; it < 16; it++) { BODY; }                             \
        long long t1 = now();                                                 \
        SINK;                                                                 \
        if (threadIdx.x == 0) out[ID] = (double)(t1 - t0) / (16.0 * (N));     \
    }

// waves 0 and 4 of a 320-thread workgroup share SIMD 0: the MFMA pipe under two waves (test 24)
__global__ __launch_bounds__(320) void k_mfma2(double* out, double* sink, double seed) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int wv = threadIdx.x >> 6;
    if (wv != 0 && wv != 4) return;
    double b = 1.0000001, c = 1e-9, a = seed + (threadIdx.x & 63) * 1e-3;
    d4 D0 = {a, a, a, a}, D1 = D0, D2 = D0, D3 = D0;
    long long t0 = now();
    for (int it = 0; it < 16; it++)
        asm volatile(R16("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\tv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\tv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\tv_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n\t")
                     : "+v"(D0), "+v"(D1), "+v"(D2), "+v"(D3) : "v"(b), "v"(c));
    long long t1 = now();
    if ((threadIdx.x & 63) == 0) out[24 + (wv ? 1 : 0)] = (double)(t1 - t0) / (16.0 * 64);
    sink[threadIdx.x] = D0[0] + D1[1] + D2[2] + D3[3];
}

__global__ __launch_bounds__(64) void k_lat(double* out, double* sink, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, acc = 0.0;
    double a0 = a, a1 = a + 1, a2 = a + 2, a3 = a + 3, a4 = a + 4, a5 = a + 5, a6 = a + 6, a7 = a + 7;
    // 0: dependent v_fma_f64
    TEST(0, 64, , asm volatile(R64("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c)), acc += a)
    // 1: 8 independent v_fma_f64 (issue rate)
    TEST(1, 64, , asm volatile(R4(R4("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t") R4("v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t") R4("") ) R4("")
         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)), acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
    // 2: dependent v_rsq_f64 (+ s_nop for the trans hazard)
    a = seed;
    TEST(2, 64, , asm volatile(R64("v_rsq_f64 %0, %0\n\ts_nop 0\n\t") : "+v"(a)), acc += a)
    // 3: dependent v_fmac_f64_dpp (acc chain)
    a = seed; 
    TEST(3, 64, , asm volatile(R64("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a) : "v"(b), "v"(c)), acc += a)
    // 4: 8 independent v_fmac_f64_dpp (issue rate)
    TEST(4, 64, , asm volatile(R4(R4("v_fmac_f64_dpp %0, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, -%8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, -%8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t") R4("v_fmac_f64_dpp %4, -%8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, -%8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, -%8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, -%8, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"))
         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)), acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
    // 5: chain readlane x2 -> v_mul_f64 with the SGPR pair (VGPR -> SGPR -> VGPR round trip + one mul)
    a = seed;
    TEST(5, 64, , asm volatile("v_mov_b64 v[100:101], %0\n\t" R64("v_readlane_b32 s20, v100, 3\n\tv_readlane_b32 s21, v101, 3\n\tv_mul_f64 v[100:101], s[20:21], %1\n\t") "v_mov_b64 %0, v[100:101]\n\t" : "+v"(a) : "v"(b) : "s20", "s21", "v100", "v101"), acc += a)
    // 6: dependent v_mul_f64 alone (compare with 5)
    a = seed;
    TEST(6, 64, , asm volatile(R64("v_mul_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b)), acc += a)
    // 7: chain of v_permlane16_swap_b32 pairs (lo, hi) on one register pair
    { unsigned x0 = threadIdx.x, x1 = threadIdx.x * 3, y0 = 5, y1 = 7;
      TEST(7, 64, , asm volatile(R64("v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\t") : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1)), acc += x0 + x1 + y0 + y1) }
    // 8: chain v_mov_b64_dpp row_newbcast (needs 2 wait states after the write: s_nop 1)
    a = seed;
    TEST(8, 64, , asm volatile(R64("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a)), acc += a)
    // 9: the pivot chain of c16_row_iter as written: fmac_dpp (even rows) -> readlane x2 -> rsq -> mul -> fma -> fma -> fma -> back
    a = seed; 
    { double y = 1.0, t = 0.0, e = 0.0, p = 0.0;
      TEST(9, 16, , asm volatile("v_mov_b64 v[100:101], %0\n\t" R16("v_fmac_f64_dpp v[100:101], -%1, %1 row_newbcast:3 row_mask:0x5 bank_mask:0xf\n\tv_readlane_b32 s20, v100, 3\n\tv_readlane_b32 s21, v101, 3\n\tv_rsq_f64 %1, s[20:21]\n\ts_nop 0\n\tv_mul_f64 %2, s[20:21], %1\n\tv_fma_f64 %3, -%2, %1, 1.0\n\tv_fma_f64 %4, %5, %3, 0.5\n\tv_fma_f64 %1, %3, %4, %1\n\ts_nop 1\n\t") "v_mov_b64 %0, v[100:101]\n\t" : "+v"(a), "+v"(y), "+v"(t), "+v"(e), "+v"(p) : "v"(b) : "s20", "s21", "v100", "v101"), acc += a + y) }
    // 10: same chain with the pivot broadcast by DPP instead of readlane (v_mov_b64_dpp -> VGPR, rsq from VGPR)
    a = seed;
    { double y = 1.0, t = 0.0, e = 0.0, p = 0.0, dv = 0.0;
      TEST(10, 16, , asm volatile(R16("v_fmac_f64_dpp %0, -%1, %1 row_newbcast:3 row_mask:0x5 bank_mask:0xf\n\ts_nop 1\n\tv_mov_b64_dpp %6, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_rsq_f64 %1, %6\n\ts_nop 0\n\tv_mul_f64 %2, %6, %1\n\tv_fma_f64 %3, -%2, %1, 1.0\n\tv_fma_f64 %4, %5, %3, 0.5\n\tv_fma_f64 %1, %3, %4, %1\n\ts_nop 1\n\t") : "+v"(a), "+v"(y), "+v"(t), "+v"(e), "+v"(p) : "v"(b), "v"(dv)), acc += a + y) }
    // 11: v_rcp_f64 dependent
    a = seed;
    TEST(11, 64, , asm volatile(R64("v_rcp_f64 %0, %0\n\ts_nop 0\n\t") : "+v"(a)), acc += a)
    // 12 / 13 / 14: dependent v_fma_f64 with EXEC = 16 lanes / 1 lane / 32 lanes (does the VALU skip passes without active lanes?)
    a = seed;
    TEST(12, 64, , asm volatile("s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0xffff\n\t" R64("v_fma_f64 %0, %0, %1, %2\n\t") "s_mov_b64 exec, s[22:23]\n\t" : "+v"(a) : "v"(b), "v"(c) : "s22", "s23"), acc += a)
    a = seed;
    TEST(13, 64, , asm volatile("s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 1\n\t" R64("v_fma_f64 %0, %0, %1, %2\n\t") "s_mov_b64 exec, s[22:23]\n\t" : "+v"(a) : "v"(b), "v"(c) : "s22", "s23"), acc += a)
    a = seed;
    TEST(14, 64, , asm volatile("s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0xffffffff\n\t" R64("v_fma_f64 %0, %0, %1, %2\n\t") "s_mov_b64 exec, s[22:23]\n\t" : "+v"(a) : "v"(b), "v"(c) : "s22", "s23"), acc += a)
    // 15: independent v_fma_f64 x8 with EXEC = 16 lanes (issue)
    TEST(15, 64, , asm volatile("s_mov_b64 s[22:23], exec\n\ts_mov_b64 exec, 0xffff\n\t" R4(R4("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t")) "s_mov_b64 exec, s[22:23]\n\t"
         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s22", "s23"), acc += a0 + a1 + a2 + a3)
    // 16: dependent v_fma_f32; 17: dependent v_pk_fma_f32
    { float fa = (float)seed, fb = 1.0000001f, fc = 1e-9f;
      TEST(16, 64, , asm volatile(R64("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(fa) : "v"(fb), "v"(fc)), acc += fa) }
    // 18: dependent MFMA f64 16x16x4 chain (D = A B + D); 19: the 4x4x4 (4 blocks) variant; 20: result -> v_mul_f64 -> both operands of the next (one pivot-step hop)
    { typedef double d4 __attribute__((ext_vector_type(4)));
      d4 D = {a, a, a, a};
      TEST(18, 16, , asm volatile(R16("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t") : "+v"(D) : "v"(b), "v"(c)), acc += D[0] + D[1] + D[2] + D[3])
      double d1 = a;
      TEST(19, 16, , asm volatile(R16("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\t") : "+v"(d1) : "v"(b), "v"(c)), acc += d1)
      D = (d4){a, a, a, a};
      double yv = 0.0;
      TEST(20, 16, , asm volatile("v_mov_b64 v[104:105], %0\n\tv_mov_b64 v[106:107], %0\n\tv_mov_b64 v[108:109], %0\n\tv_mov_b64 v[110:111], %0\n\t"
                                  R16("v_mul_f64 %0, v[106:107], %1\n\tv_mfma_f64_16x16x4_f64 v[104:111], %0, %0, v[104:111]\n\t") "s_nop 7\n\ts_nop 7\n\tv_mov_b64 %0, v[106:107]\n\t"
                                  : "+v"(yv) : "v"(b) : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"), acc += yv) }
    // 23: four INDEPENDENT accumulators of v_mfma_f64_16x16x4 from one wave (pipe occupancy per instruction)
    { typedef double d4 __attribute__((ext_vector_type(4)));
      d4 D0 = {a, a, a, a}, D1 = D0, D2 = D0, D3 = D0;
      TEST(23, 64, , asm volatile(R16("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\tv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\tv_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\tv_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n\t")
                                  : "+v"(D0), "+v"(D1), "+v"(D2), "+v"(D3) : "v"(b), "v"(c)), acc += D0[0] + D1[1] + D2[2] + D3[3]) }
    // 21: wave-private LDS round trip: ds_write_b64 then ds_read_b64 of another lane's slot, dependent chain
    { __shared__ double lbuf[64];
      a = seed;
      TEST(21, 16, , for (int q = 0; q < 16; q++) { lbuf[threadIdx.x] = a; asm volatile("" ::: "memory"); a = lbuf[threadIdx.x ^ 17] * b; asm volatile("" ::: "memory"); }, acc += a) }
    // 22: ds_bpermute_b32 x2 (64-bit shuffle) dependent chain
    a = seed;
    TEST(22, 16, , for (int q = 0; q < 16; q++) { a = __shfl(a, (threadIdx.x + 17) & 63) * b; }, acc += a)
    sink[threadIdx.x] = acc;
}

int main() {
    CK(hipSetDevice(0));
    double *out, *sink;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&sink, 512 * 8));
    CK(hipMemset(out, 0, 64 * 8));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, sink, 1.25);
        hipLaunchKernelGGL(k_mfma2, dim3(1), dim3(320), 0, 0, out, sink, 1.25);
        CK(hipDeviceSynchronize());
    }
    double h[64];
    CK(hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost));
    const char* names[] = {"dependent v_fma_f64", "independent v_fma_f64 (issue)", "dependent v_rsq_f64 (+s_nop 0)", "dependent v_fmac_f64_dpp",
                           "independent v_fmac_f64_dpp (issue)", "readlane x2 -> v_mul_f64 with the SGPR pair (round trip)", "dependent v_mul_f64",
                           "v_permlane16_swap_b32 x2 chain", "s_nop 1 + v_mov_b64_dpp chain", "pivot chain via readlane (per column)",
                           "pivot chain via v_mov_b64_dpp (per column)", "dependent v_rcp_f64 (+s_nop 0)",
                           "dependent v_fma_f64, EXEC = 16 lanes", "dependent v_fma_f64, EXEC = 1 lane", "dependent v_fma_f64, EXEC = 32 lanes", "independent v_fma_f64, EXEC = 16 lanes (issue)",
                           "dependent v_fma_f32", "(unused)", "dependent v_mfma_f64_16x16x4 (accumulator chain)", "dependent v_mfma_f64_4x4x4 (accumulator chain)",
                           "mfma 16x16x4 -> v_mul_f64 -> mfma operand (one hop)", "LDS round trip: ds_write_b64 -> ds_read_b64 -> v_mul_f64", "ds_bpermute x2 -> v_mul_f64",
                           "independent v_mfma_f64_16x16x4 (4 accumulators, one wave), per instruction", "same, two waves on one SIMD: wave 0", "same, two waves on one SIMD: wave 4"};
    for (int i = 0; i < 26; i++) printf("%-62s %7.1f cycles (s_memtime)\n", names[i], h[i]);
    return 0;
}
