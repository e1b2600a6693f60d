// tools/experiments/lds_phase_bench.cpp — which lanes of a wave does gfx950 serve TOGETHER in one pass of a 16-byte (and 8-byte) LDS access?
// Layout decisions (pitch, swizzle) depend on it and the guides only say "16-lane groups over 64 banks".
//
// Method (sixteen waves of one workgroup issue the same access so that the LDS pipeline, not the latency, is timed): every lane l of a wave reads 16 bytes at l * 16 (conflict-free by any model: 64 lanes x 4 banks = 4 passes over 64 banks),
// except lane j, which is moved onto lane 0's banks in another row (+ 4 KB).  If lanes 0 and j are served in the same pass the access
// takes one more pass; otherwise nothing changes.  The clock of 4096 dependent-free reads per j in 1..63 gives the pass partition.  The same
// for ds_read_b64 and ds_write_b128, then the access patterns of the product kernels (conv3 / conv_out halo pitch 144, swizzled 128-byte rows).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/lds_phase_bench.cpp -o tools/lds_phase_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

// mode 0: ds_read_b128, 1: ds_read_b64, 2: ds_write_b128
template <int MODE>
__global__ __launch_bounds__(1024) void k(const int* __restrict__ offs, long long* __restrict__ clocks, int* __restrict__ sink, int npat) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;                  // sixteen waves issue the SAME access: the LDS pipeline is saturated, time = passes
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 1024) ((int*)lds)[i] = i;
    __syncthreads();
    int acc = 0;
    for (int p = 0; p < npat; ++p) {
        const int off = offs[p * 64 + lane];
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (off >= 0) {                                     // off < 0: the lane sits the pattern out (EXEC off for every access)
        // eight accesses per asm block, no consumer instructions: the loop is bound by the LDS pipeline, not by VALU issue or latency
        if constexpr (MODE == 0) {
            i32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            for (int it = 0; it < ITERS / 8; ++it) {
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8\n ds_read_b128 %2, %8\n ds_read_b128 %3, %8\n"
                             "ds_read_b128 %4, %8\n ds_read_b128 %5, %8\n ds_read_b128 %6, %8\n ds_read_b128 %7, %8\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(off) : "memory");
            }
            acc ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3] ^ v4[0] ^ v5[1] ^ v6[2] ^ v7[3];
        } else if constexpr (MODE == 1) {
            i32x2 v0, v1, v2, v3, v4, v5, v6, v7;
            for (int it = 0; it < ITERS / 8; ++it) {
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                             "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(off) : "memory");
            }
            acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
        } else {
            const i32x4 v = {lane, p, lane, p};
            for (int it = 0; it < ITERS / 8; ++it) {
                asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n"
                             "ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n"
                             : : "v"(off), "v"(v) : "memory");
            }
        }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) clocks[p] = t1 - t0;
    }
    sink[threadIdx.x & 63] = acc;
}

template <int MODE>
static std::vector<double> run(const std::vector<int>& offs) {
    const int npat = (int)offs.size() / 64;
    int* doffs; long long* dclk; int* dsink;
    CK(hipMalloc(&doffs, offs.size() * 4)); CK(hipMalloc(&dclk, npat * 8)); CK(hipMalloc(&dsink, 256));
    CK(hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(1024), 0, 0, doffs, dclk, dsink, npat); CK(hipDeviceSynchronize()); }
    std::vector<long long> c(npat); CK(hipMemcpy(c.data(), dclk, npat * 8, hipMemcpyDeviceToHost));
    std::vector<double> r(npat); for (int i = 0; i < npat; ++i) r[i] = (double)c[i] / ITERS / 16.0;     // per wave-access
    hipFree(doffs); hipFree(dclk); hipFree(dsink);
    return r;
}

// pass partition: only lanes 0 and j are active.  Pattern 2j: lane j on lane 0's banks in another row; pattern 2j+1: lane j on other banks.
// The difference is a pass that had to be split, i.e. lanes 0 and j are served together.
template <int MODE>
static void partition(const char* what) {
    std::vector<int> offs;
    for (int j = 1; j < 64; ++j) {
        for (int l = 0; l < 64; ++l) offs.push_back(l == 0 ? 0 : (l == j ? 4096 : -1));
        for (int l = 0; l < 64; ++l) offs.push_back(l == 0 ? 0 : (l == j ? 4096 + 128 : -1));
    }
    const std::vector<double> r = run<MODE>(offs);
    printf("== %s, two active lanes (0, j): clocks per access with j on OTHER banks %.2f; lanes j that share a pass with lane 0 (same banks cost more):\n   ", what, r[1]);
    for (int j = 1; j < 64; ++j) if (r[2 * (j - 1)] > r[2 * (j - 1) + 1] * 1.2) printf("%d(%.2f) ", j, r[2 * (j - 1)]);
    printf("\n");
}

// what does an out-of-range lane of buffer_load_dwordx4 ... lds leave in LDS: zeros (like the VGPR form) or nothing?
__global__ __launch_bounds__(64) void dma_oor_kernel(const int* __restrict__ src, int nbytes, int* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) int lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0x7f7f7f7f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int voff = lane < 32 ? lane * 16 : nbytes;             // lanes 32..63 out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

static void dma_oor() {
    std::vector<int> h(256); for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    int *dsrc, *dout; CK(hipMalloc(&dsrc, 1024)); CK(hipMalloc(&dout, 1024));
    CK(hipMemcpy(dsrc, h.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dma_oor_kernel, dim3(1), dim3(64), 0, 0, dsrc, 1024, dout); CK(hipDeviceSynchronize());
    std::vector<int> o(256); CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
    int ok_lo = 0, zero_hi = 0, sentinel_hi = 0;
    for (int i = 0; i < 128; ++i) ok_lo += o[i] == 1000 + i;
    for (int i = 128; i < 256; ++i) { zero_hi += o[i] == 0; sentinel_hi += o[i] == 0x7f7f7f7f; }
    printf("== buffer_load_dwordx4 ... lds with lanes 32..63 out of range: in-range dwords correct %d / 128; out-of-range slots: %d zero, %d untouched (of 128), first %08x\n",
           ok_lo, zero_hi, sentinel_hi, (unsigned)o[128]);
    hipFree(dsrc); hipFree(dout);
}

int main() {
    dma_oor();
    partition<0>("ds_read_b128");
    partition<1>("ds_read_b64");
    partition<2>("ds_write_b128");
    // access patterns of the product kernels, ds_read_b128
    struct Pat { const char* name; int (*f)(int); };
    const Pat pats[] = {
        {"contiguous l*16", [](int l) { return l * 16; }},
        {"conv_out halo: (l%16)*144 + (l/16)*16", [](int l) { return (l % 16) * 144 + (l / 16) * 16; }},
        {"conv3 C=64 halo: (l%32)*144 + (l/32)*16", [](int l) { return (l % 32) * 144 + (l / 32) * 16; }},
        {"conv3 C=128 halo: (l%32)*272 + (l/32)*16", [](int l) { return (l % 32) * 272 + (l / 32) * 16; }},
        {"stage image 128-byte rows, swizzle (row>>1)&7", [](int l) { const int r = l % 32; return r * 128 + (((l / 32) ^ ((r >> 1) & 7)) << 4); }},
        {"128-byte rows, no swizzle", [](int l) { return (l % 32) * 128 + (l / 32) * 16; }},
        {"64-byte rows, swizzle (row>>2)&3", [](int l) { const int r = l % 32; return r * 64 + (((l / 32) ^ ((r >> 2) & 3)) << 4); }},
        {"conv_out with pitch 80: (l%16)*80 + (l/16)*16", [](int l) { return (l % 16) * 80 + (l / 16) * 16; }},
        {"conv_out with 128-byte pixels, swizzle (px>>1)&7", [](int l) { const int r = l % 16; return r * 128 + (((l / 16) ^ ((r >> 1) & 7)) << 4); }},
        {"conv_out with 128-byte pixels, swizzle px&7", [](int l) { const int r = l % 16; return r * 128 + (((l / 16) ^ (r & 7)) << 4); }},
        {"all lanes one address (broadcast)", [](int) { return 0; }},
    };
    std::vector<int> offs;
    for (const Pat& p : pats) for (int l = 0; l < 64; ++l) offs.push_back(p.f(l));
    const std::vector<double> r = run<0>(offs), w = run<2>(offs);
    printf("== patterns (clocks per access: ds_read_b128 / ds_write_b128)\n");
    for (size_t i = 0; i < sizeof(pats) / sizeof(pats[0]); ++i) printf("   %-52s %6.2f / %6.2f\n", pats[i].name, r[i], w[i]);
    return 0;
}
