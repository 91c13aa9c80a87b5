// What a hand-off INSIDE one persistent launch costs on this chip, against the kernel boundary it would replace
// ("if the grid barrier measures > 3 us, commit launch_floor-style evidence").  Not part of the product.
//
// One workgroup per CU (256 threads), `phases` phases.  A phase is what a fused low-resolution level would do between two
// layers: every workgroup reads the `bytes / grid` block its NEIGHBOUR (workgroup + shift) stored in the previous phase,
// adds one and stores its own block; then all workgroups meet.  Two barriers:
//   flat  - one monotonic device-scope counter: lane 0 release fence -> atomic add -> relaxed sc1 poll (+ s_sleep) -> acquire fence
//   xcd   - hierarchical (MI355X_MICROARCH.md "barrier-xcd"): per-XCC arrival counter; the LAST arriver of an XCC releases,
//           adds to the top counter, polls it, acquires and bumps the XCC's generation word; the others poll that word
//           (served by their own XCD's L2) and acquire.
// Every poll is bounded: a barrier that cannot complete (workgroups not co-resident) raises err[0] and the launch ends.
// The same phase body as `phases` dependent launches of one graph is tools/launch_floor.hip's launch_chain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define POLL_LIMIT (1u << 20)

struct Sync {
    unsigned top;          // flat counter
    unsigned pad0[31];
    unsigned top2;         // top counter of the hierarchy (its own line)
    unsigned pad1[31];
    unsigned xcc_arrive[8 * 32];   // one 128-byte line per XCC
    unsigned xcc_gen[8 * 32];
    unsigned xcc_pop[8 * 32];      // census: workgroups resident per XCC
    unsigned err;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ bool poll_ge(unsigned* p, unsigned target, unsigned* err) {
    unsigned n = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > POLL_LIMIT) { atomicOr(err, 1u); return false; }
    }
    return true;
}

// all workgroups; `epoch` counts barriers from 1
__device__ __forceinline__ void barrier_flat(Sync* s, unsigned epoch, unsigned grid) {
    __syncthreads();   // (every storing wave has waited vmcnt(0): the workgroup-scope release of __syncthreads)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poll_ge(&s->top, epoch * grid, &s->err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(Sync* s, unsigned epoch, unsigned xcc, unsigned pop, unsigned nxcc) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&s->xcc_arrive[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == epoch * pop) {   // last arriver of this XCC
            __hip_atomic_fetch_add(&s->top2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            poll_ge(&s->top2, epoch * nxcc, &s->err);
            __hip_atomic_store(&s->xcc_gen[xcc * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            poll_ge(&s->xcc_gen[xcc * 32], epoch, &s->err);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// mode 0: flat, 1: xcd-hierarchical, 2: NO barrier (the phase bodies alone: wrong results, the time of the work itself)
template <int MODE>
__global__ __launch_bounds__(256) void k_phases(int4* a, int4* b, int n16_per_wg, int phases, int shift, Sync* s) {
    const unsigned grid = gridDim.x, wg = blockIdx.x;
    unsigned xcc = 0, pop = 0, nxcc = 0, epoch = 0;
    if (MODE == 1) {   // census, then one flat barrier so that every workgroup reads final populations
        xcc = xcc_id();
        if (threadIdx.x == 0) atomicAdd(&s->xcc_pop[xcc * 32], 1u);
        barrier_flat(s, 1, grid);
        pop = __hip_atomic_load(&s->xcc_pop[xcc * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int x = 0; x < 8; ++x) nxcc += __hip_atomic_load(&s->xcc_pop[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    const unsigned src_wg = (wg + (unsigned)shift) % grid;
    for (int p = 0; p < phases; ++p) {
        const int4* src = ((p & 1) ? b : a) + (size_t)src_wg * n16_per_wg;
        int4* dst = ((p & 1) ? a : b) + (size_t)wg * n16_per_wg;
        for (int i = threadIdx.x; i < n16_per_wg; i += 256) {
            int4 v = src[i];
            v.x += 1; v.y += 1; v.z += 1; v.w += 1;
            dst[i] = v;
        }
        ++epoch;
        if (MODE == 0) barrier_flat(s, epoch, grid);
        if (MODE == 1) barrier_xcd(s, epoch, xcc, pop, nxcc);
        if (MODE == 2) __syncthreads();
        if (MODE != 2) {   // a barrier that gave up ends the launch for everybody: no second wait
            __shared__ unsigned bad;
            if (threadIdx.x == 0) bad = __hip_atomic_load(&s->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (bad) return;
        }
    }
}

extern "C" int launch_phases(void* a, void* b, int n16_per_wg, int phases, int shift, int grid, int mode, void* sync, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_phases<0>, dim3(grid), dim3(256), 0, st, (int4*)a, (int4*)b, n16_per_wg, phases, shift, (Sync*)sync);
    else if (mode == 1) hipLaunchKernelGGL(k_phases<1>, dim3(grid), dim3(256), 0, st, (int4*)a, (int4*)b, n16_per_wg, phases, shift, (Sync*)sync);
    else hipLaunchKernelGGL(k_phases<2>, dim3(grid), dim3(256), 0, st, (int4*)a, (int4*)b, n16_per_wg, phases, shift, (Sync*)sync);
    return (int)hipGetLastError();
}

// the same phase body as its own launch (one per phase, dependent through a / b)
__global__ __launch_bounds__(256) void k_one_phase(const int4* src_base, int4* dst_base, int n16_per_wg, int shift) {
    const unsigned grid = gridDim.x, wg = blockIdx.x;
    const int4* src = src_base + (size_t)((wg + (unsigned)shift) % grid) * n16_per_wg;
    int4* dst = dst_base + (size_t)wg * n16_per_wg;
    for (int i = threadIdx.x; i < n16_per_wg; i += 256) {
        int4 v = src[i];
        v.x += 1; v.y += 1; v.z += 1; v.w += 1;
        dst[i] = v;
    }
}

extern "C" int launch_one_phase(const void* src, void* dst, int n16_per_wg, int shift, int grid, void* stream) {
    hipLaunchKernelGGL(k_one_phase, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const int4*)src, (int4*)dst, n16_per_wg, shift);
    return (int)hipGetLastError();
}

extern "C" int sync_bytes() { return (int)sizeof(Sync); }
