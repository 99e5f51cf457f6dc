// lds_bcast.hip — how many LDS-pipe cycles a broadcast ds_read_b128 (every lane the same address: the filter's operand
// fetch) costs a CU, next to a ds_read_b128 with a different address per lane, a broadcast ds_read_b64 and ds_read_b32.
// Each wave issues `iters` rounds of 16 reads (offsets 0..240 from its base) and waits once per round; no VALU work
// besides the loop counter.  `waves` waves per SIMD on every CU (one block per CU, forced by the LDS allocation).
// Output: elapsed shader cycles of the slowest wave / reads issued by the CU = LDS cycles per wave-level read.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_bcast.hip -o tools/bin/lds_bcast && tools/bin/lds_bcast 5 2000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define R16(INS, T, OUT)                                                                                                  \
    asm volatile(INS " %0, %16\n" INS " %1, %16 offset:16\n" INS " %2, %16 offset:32\n" INS " %3, %16 offset:48\n"        \
                 INS " %4, %16 offset:64\n" INS " %5, %16 offset:80\n" INS " %6, %16 offset:96\n" INS " %7, %16 offset:112\n" \
                 INS " %8, %16 offset:128\n" INS " %9, %16 offset:144\n" INS " %10, %16 offset:160\n" INS " %11, %16 offset:176\n" \
                 INS " %12, %16 offset:192\n" INS " %13, %16 offset:208\n" INS " %14, %16 offset:224\n" INS " %15, %16 offset:240\n" \
                 "s_waitcnt lgkmcnt(0)"                                                                                   \
                 : "=v"(OUT[0]), "=v"(OUT[1]), "=v"(OUT[2]), "=v"(OUT[3]), "=v"(OUT[4]), "=v"(OUT[5]), "=v"(OUT[6]), "=v"(OUT[7]),      \
                   "=v"(OUT[8]), "=v"(OUT[9]), "=v"(OUT[10]), "=v"(OUT[11]), "=v"(OUT[12]), "=v"(OUT[13]), "=v"(OUT[14]), "=v"(OUT[15]) \
                 : "v"(addr))

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cycles, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.5f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // MODE 0/2/3: every lane the same address (broadcast); MODE 1: 16 bytes per lane, consecutive (conflict-free)
    unsigned addr = MODE == 1 ? (wave * 1024 + lane * 16) : wave * 1024;
    f4 v[16]; f2 w[16]; float s[16];
    float acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 1) { R16("ds_read_b128", f4, v); acc += v[0].x + v[15].w; }
        if (MODE == 2) { R16("ds_read_b64", f2, w); acc += w[0].x + w[15].y; }
        if (MODE == 3) { R16("ds_read_b32", float, s); acc += s[0] + s[15]; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves, int iters, int n_cu, float* d_out, unsigned long long* d_cyc) {
    const int threads = 256 * waves, lds_bytes = 96 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_cyc, 10);
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu), dim3(threads), lds_bytes, 0, d_out, d_cyc, iters);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(static_cast<size_t>(n_cu) * waves * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (auto c : h) mx = c > mx ? static_cast<double>(c) : mx;
    const double reads_per_cu = static_cast<double>(iters) * 16 * waves * 4;
    std::printf("%-34s %d waves/SIMD: %.2f LDS cycles per wave-level read (slowest wave %.0f cycles, %.0f reads per CU)\n", name, waves, mx / reads_per_cu, mx, reads_per_cu);
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? std::atoi(argv[1]) : 4, iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    if (waves < 1 || waves > 4 || iters < 1 || iters > 100000) { std::fprintf(stderr, "usage: lds_bcast [waves per SIMD 1..4] [rounds]\n"); return 2; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float* d_out; unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(n_cu) * 1024 * 4));
    CHECK(hipMalloc(&d_cyc, static_cast<size_t>(n_cu) * 16 * 8));
    run<0>("ds_read_b128, broadcast", waves, iters, n_cu, d_out, d_cyc);
    run<1>("ds_read_b128, 16 B per lane", waves, iters, n_cu, d_out, d_cyc);
    run<2>("ds_read_b64, broadcast", waves, iters, n_cu, d_out, d_cyc);
    run<3>("ds_read_b32, broadcast", waves, iters, n_cu, d_out, d_cyc);
    return 0;
}
