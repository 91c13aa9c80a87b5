// Is the instruction cache cold at every kernel launch, and what does straight-line cold code cost?
//   hipcc --offload-arch=gfx950 -O2 tools/icache_cold_probe.hip -o tools/_build/ic_probe && tools/_build/ic_probe   (stand-alone, off the product path)
// Measured (round 4, MI355X): 8 KB of straight-line VALU code, hot 5,328 clocks; first launch ever +2,450; every later eager launch
// +500-600 (0.22-0.25 us) whether or not another kernel ran in between; inside a replayed graph of the same kernel +70.
// Kernel: a block of NI distinct VALU instructions (about 8 bytes each) executed twice in a loop; thread 0 of every workgroup stamps
// s_memtime around each pass.  Launched several times back to back (and alternating with a second kernel of different code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define R256(x) R64(x) R64(x) R64(x) R64(x)
template <int VAR>
__global__ __launch_bounds__(256) void body(unsigned long long* out, float* sink, int passes) {
    float a = threadIdx.x * 0.5f + VAR, b = 1.0001f;
    unsigned long long t[5];
    t[0] = stamp();
    for (int p = 0; p < passes && p < 4; ++p) {
        // 1024 dependent-free-ish VALU instructions with 64-bit encodings (literal constants): ~8 KB of code
        R256(asm volatile("v_fmac_f32 %0, 0x3f800347, %1\n\tv_add_f32 %1, 0x3a83126f, %1\n\tv_fmac_f32 %0, 0x3f7fbe77, %1\n\tv_add_f32 %1, 0xba83126f, %1" : "+v"(a), "+v"(b));)
        t[p + 1] = stamp();
    }
    if (threadIdx.x == 0)
        for (int p = 0; p < 5; ++p) out[blockIdx.x * 8 + p] = t[p];
    if (a == 12345.678f) sink[0] = a + b;
}
int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8 * 256 * sizeof(unsigned long long)); hipMalloc(&s, 64);
    std::vector<unsigned long long> h(8 * 256);
    auto report = [&](const char* name) {
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> p1, p2, p3;
        for (int w = 0; w < 256; ++w) { p1.push_back((double)(h[w*8+1]-h[w*8+0])); p2.push_back((double)(h[w*8+2]-h[w*8+1])); p3.push_back((double)(h[w*8+3]-h[w*8+2])); }
        std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end()); std::sort(p3.begin(), p3.end());
        printf("%-44s pass1 %7.0f / %7.0f / %7.0f   pass2 %7.0f / %7.0f   pass3 %7.0f  (min / median / max ticks of 10 ns)\n", name, p1[0], p1[128], p1[255], p2[0], p2[128], p3[128]);
    };
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(body<0>, dim3(256), dim3(256), 0, 0, d, s, 3); hipDeviceSynchronize(); report(rep == 0 ? "A first launch ever" : "A again (after sync)"); }
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(body<0>, dim3(256), dim3(256), 0, 0, d, s, 3);
    hipDeviceSynchronize(); report("A x4 back to back (last)");
    hipLaunchKernelGGL(body<1>, dim3(256), dim3(256), 0, 0, d, s, 3); hipDeviceSynchronize(); report("B first");
    for (int rep = 0; rep < 4; ++rep) { hipLaunchKernelGGL(body<0>, dim3(256), dim3(256), 0, 0, d, s, 3); hipLaunchKernelGGL(body<1>, dim3(256), dim3(256), 0, 0, d, s, 3); }
    hipDeviceSynchronize(); report("ABAB... (last = B)");
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(body<1>, dim3(256), dim3(256), 0, 0, d, s, 3);
    hipDeviceSynchronize(); report("BBBB (last)");
    // graph replay of AAAA
    hipStream_t st; hipStreamCreate(&st); hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int rep = 0; rep < 6; ++rep) hipLaunchKernelGGL(body<0>, dim3(256), dim3(256), 0, st, d, s, 3);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st); hipGraphLaunch(ge, st); hipStreamSynchronize(st); report("graph of 6 x A (last)");
    return 0;
}
