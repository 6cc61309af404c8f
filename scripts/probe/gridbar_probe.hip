// gridbar_probe.hip — cost of an in-kernel barrier across all workgroups of a launch on MI355X (8 XCDs, one L2 each): every
// workgroup writes a record, release-increments ONE counter, thread 0 spins on it (acquire), then all threads read every record.
// Reported: time per barrier round (kernel of R rounds, hipEvents), for 64 / 258 / 512 workgroups, with the counter polled by
// relaxed atomic loads + one fence, and the wall-clock spread between the first and the last workgroup leaving a round.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probe/_build/gridbar_probe scripts/probe/gridbar_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ long long wall() { long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }   // 100 MHz

template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, double* rec, double* out, long long* stamps, int rounds, int payload) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int r = 0; r < rounds; r++) {
        // payload: this workgroup's record (+ some dirty lines to write back)
        for (int i = tid; i < payload; i += 256) rec[((size_t)r & 1) * nb * payload + (size_t)b * payload + i] = b + r + i * 1e-3;
        __syncthreads();
        if (tid == 0) {
            if (MODE == 0) {
                __hip_atomic_fetch_add(ctr + r, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb) __builtin_amdgcn_s_sleep(1);
                __atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default for __atomic_thread_fence in HIP? use the scoped builtin below
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(ctr + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb) ;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            if (r == rounds - 1) stamps[b] = wall();
        }
        __syncthreads();
        // consume: every thread reads the first word of up to 4 records (like the tile partials of k_build)
        const double* rr = rec + ((size_t)r & 1) * nb * payload;
        for (int u = 0; u < 4; u++) { const int t = tid + u * 256; if (t < nb) acc += __builtin_nontemporal_load(rr + (size_t)t * payload); }
    }
    if (tid == 0) out[b] = acc;
    else if (acc == 12345.678) out[b] = acc;
}

int main() {
    CK(hipSetDevice(0));
    const int rounds = 20;
    unsigned* ctr; double *rec, *out; long long* st;
    CK(hipMalloc(&ctr, 4 * 64)); CK(hipMalloc(&rec, 8ull * 2 * 512 * 4096)); CK(hipMalloc(&out, 8 * 512)); CK(hipMalloc(&st, 8 * 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++)
        for (int payload : {8, 512, 4096})
            for (int nb : {64, 258, 512}) {
                float best = 1e9f; double sp = 0;
                for (int rep = 0; rep < 5; rep++) {
                    CK(hipMemset(ctr, 0, 4 * 64));
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(nb), dim3(256), 0, 0, ctr, rec, out, st, rounds, payload);
                    else hipLaunchKernelGGL(k_bar<1>, dim3(nb), dim3(256), 0, 0, ctr, rec, out, st, rounds, payload);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) {
                        best = ms;
                        std::vector<long long> h(nb); CK(hipMemcpy(h.data(), st, 8 * nb, hipMemcpyDeviceToHost));
                        long long lo = h[0], hi = h[0]; for (auto v : h) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
                        sp = (hi - lo) * 0.01;
                    }
                }
                // check
                std::vector<double> o(nb); CK(hipMemcpy(o.data(), out, 8 * nb, hipMemcpyDeviceToHost));
                double expect = 0; for (int r = 0; r < rounds; r++) for (int t = 0; t < nb && t < 1024; t += 256) expect += t + r;   // thread 0 reads records 0, 256, 512, 768
                printf("mode %d (%s) payload %4d doubles, %3d workgroups: %.2f us per round (kernel of %d rounds incl. launch: %.1f us), exit spread of the last round %.2f us, result %s\n",
                       mode, mode ? "fence + relaxed add, tight spin" : "release add, s_sleep spin", payload, nb, 1e3 * best / rounds, rounds, 1e3 * best, sp, o[0] == expect ? "ok" : "WRONG");
            }
    return 0;
}
