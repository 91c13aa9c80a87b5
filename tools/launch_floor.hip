// Micro-probe (not part of the product): what one kernel boundary costs inside a replayed hipGraph on this box.
//   k_empty: nothing.   k_chain: every thread loads 16 B of what the PREVIOUS launch stored (ping-pong buffers), adds, stores 16 B —
//   what its own workgroup stored (same XCD under round-robin placement) or what workgroup + shift stored (another XCD).
// tools/launch_floor.py captures N launches into a graph and reports the per-launch time for grids of 1 / 256 / 1024 workgroups
// and chain footprints of 64 KB / 1 MB / 8 MB: launch + drain + the cross-XCD round trip of a dependent chain's first load.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void k_empty() {}
// shift: the workgroup reads what workgroup (blockIdx.x + shift) of the previous launch stored (shift % 8 != 0: another XCD's stores)
extern "C" __global__ __launch_bounds__(256) void k_chain(const uint4* in, uint4* out, int n16, int shift) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        int j = i + shift * 256;
        if (j >= n16) j -= n16;
        uint4 v = in[j];
        v.x += 1; v.y += v.x; v.z ^= v.y; v.w += 3;
        out[i] = v;
    }
}
extern "C" int launch_empty(int grid, void* stream) {
    hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, (hipStream_t)stream);
    return (int)hipGetLastError();
}
extern "C" int launch_chain(const void* in, void* out, int n16, int grid, int shift, void* stream) {
    hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, n16, shift);
    return (int)hipGetLastError();
}
