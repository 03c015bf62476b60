// probe_fetch_calib.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the model's
// kernels use (the guide's "FETCH_SIZE reports half the bytes of a wide coalesced streaming read" is
// calibrated on plain 16-B-per-lane loads; the GEMMs read through LDS-DMA, the full-row kernel's A slabs in 128-B row pieces,
// the bf16 residual in 8-B lanes).  Every kernel moves a KNOWN number of bytes of a 1 GiB buffer (4x the Infinity Cache, each
// byte touched once per launch); compare with the counters:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_fetch tools/probe_fetch_calib.hip
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- /tmp/probe_fetch
//     rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -o w -- /tmp/probe_fetch      (tools/fetch_calib_report.py sums up)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr size_t BYTES = 1ull << 30;
constexpr int WGS = 2048, THREADS = 256;

// (1) plain wide streaming read: 16 B per lane, a wave covers 1 KiB contiguous (the guide's calibration case)
__global__ __launch_bounds__(THREADS) void read_x4(const u32x4* __restrict__ src, unsigned* __restrict__ sink, size_t n16) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n16; i += (size_t)WGS * THREADS) acc ^= src[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1;
}
// (2) 8 B per lane, a wave covers 512 B contiguous
__global__ __launch_bounds__(THREADS) void read_x2(const u32x2* __restrict__ src, unsigned* __restrict__ sink, size_t n8) {
    u32x2 acc = {0, 0};
    for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n8; i += (size_t)WGS * THREADS) acc ^= src[i];
    if ((acc[0] ^ acc[1]) == 0x12345678u) sink[blockIdx.x] = 1;
}
// (3) the bf16 residual read of gemm_frd (HB): lane (r32, hh) reads 8 B at row r32, byte 16 g + 8 hh of a 1536-B row: a wave
//     instruction touches 32 rows x 16 B; the four g of a group make 64 B per row, two groups a whole line.  Rows of 1536 B.
__global__ __launch_bounds__(THREADS) void read_rows_x2(const char* __restrict__ src, unsigned* __restrict__ sink, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
    u32x2 acc = {0, 0};
    for (size_t tile = blockIdx.x; tile * 128 < nrows; tile += WGS) {          // 128 rows per workgroup pass, wave w: 192 columns (384 B)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const char* rp = src + (tile * 128 + mb * 32 + r32) * 1536 + wave * 384 + 8 * hh;
#pragma unroll
            for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc ^= *reinterpret_cast<const u32x2*>(rp + nb * 64 + g * 16);
        }
    }
    if ((acc[0] ^ acc[1]) == 0x12345678u) sink[blockIdx.x] = 1;
}
// (4) the same rows as fp32 (gemm_frd's fp32 residual): 16 B per lane at row r32, byte 32 g + 16 hh of a 3072-B row
__global__ __launch_bounds__(THREADS) void read_rows_x4(const char* __restrict__ src, unsigned* __restrict__ sink, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t tile = blockIdx.x; tile * 128 < nrows; tile += WGS) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const char* rp = src + (tile * 128 + mb * 32 + r32) * 3072 + wave * 768 + 16 * hh;
#pragma unroll
            for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc ^= *reinterpret_cast<const u32x4*>(rp + nb * 128 + g * 32);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1;
}
// (5) LDS-DMA, wave-contiguous: one global_load_lds_dwordx4 = 1 KiB contiguous per wave (the W stream of gemm_fr / gemm256's
//     K-contiguous tiles are 128-B row pieces, see (6))
__global__ __launch_bounds__(THREADS) void dma_linear(const char* __restrict__ src, unsigned* __restrict__ sink, size_t n1k) {
    __shared__ __attribute__((aligned(16))) char buf[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)buf[wave]);
    for (size_t i = blockIdx.x * 4 + wave; i < n1k; i += (size_t)WGS * 4) {
        const char* g = src + i * 1024 + lane * 16;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (buf[wave][lane] == 0x77 && buf[wave][lane + 64] == 0x78) sink[blockIdx.x] = 1;
}
// (6) LDS-DMA in 128-B row pieces: a wave instruction = 8 rows x 128 B of a K-contiguous operand (row stride 1536 B): the A / W
//     tiles of the tiled GEMMs and the A slabs of the full-row kernels
__global__ __launch_bounds__(THREADS) void dma_rows(const char* __restrict__ src, unsigned* __restrict__ sink, size_t nrows) {
    __shared__ __attribute__((aligned(16))) char buf[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, arow = lane >> 3, apos = lane & 7;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr_t)buf[wave]);
    for (size_t tile = blockIdx.x; tile * 32 < nrows; tile += WGS) {           // 32 rows per workgroup pass (8 per wave), all 12 slabs of 128 B
        const char* rp = src + (tile * 32 + wave * 8 + arow) * 1536 + apos * 16;
#pragma unroll
        for (int sl = 0; sl < 12; ++sl) {
            const char* g = rp + sl * 128;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (buf[wave][lane] == 0x77 && buf[wave][lane + 64] == 0x78) sink[blockIdx.x] = 1;
}
// (7) / (8) streaming stores, 16 B and 8 B per lane
__global__ __launch_bounds__(THREADS) void write_x4(u32x4* __restrict__ dst, size_t n16) {
    const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n16; i += (size_t)WGS * THREADS) dst[i] = v;
}
__global__ __launch_bounds__(THREADS) void write_x2(u32x2* __restrict__ dst, size_t n8) {
    const u32x2 v = {1u, (unsigned)blockIdx.x};
    for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n8; i += (size_t)WGS * THREADS) dst[i] = v;
}
// (9) non-temporal 16-B stores (the fp32 h stores of the full-row kernels)
__global__ __launch_bounds__(THREADS) void write_x4_nt(u32x4* __restrict__ dst, size_t n16) {
    const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < n16; i += (size_t)WGS * THREADS)
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst + i), "v"(v) : "memory");
}

int main() {
    char* buf = nullptr;
    unsigned* sink = nullptr;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&sink, WGS * 4));
    CK(hipMemset(buf, 0x5a, BYTES));
    CK(hipMemset(sink, 0, WGS * 4));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_x4, dim3(WGS), dim3(THREADS), 0, 0, (const u32x4*)buf, sink, BYTES / 16);
        hipLaunchKernelGGL(read_x2, dim3(WGS), dim3(THREADS), 0, 0, (const u32x2*)buf, sink, BYTES / 8);
        hipLaunchKernelGGL(read_rows_x2, dim3(WGS), dim3(THREADS), 0, 0, (const char*)buf, sink, BYTES / 1536 / 128 * 128);
        hipLaunchKernelGGL(read_rows_x4, dim3(WGS), dim3(THREADS), 0, 0, (const char*)buf, sink, BYTES / 3072 / 128 * 128);
        hipLaunchKernelGGL(dma_linear, dim3(WGS), dim3(THREADS), 0, 0, (const char*)buf, sink, BYTES / 1024);
        hipLaunchKernelGGL(dma_rows, dim3(WGS), dim3(THREADS), 0, 0, (const char*)buf, sink, BYTES / 1536 / 32 * 32);
        hipLaunchKernelGGL(write_x4, dim3(WGS), dim3(THREADS), 0, 0, (u32x4*)buf, BYTES / 16);
        hipLaunchKernelGGL(write_x2, dim3(WGS), dim3(THREADS), 0, 0, (u32x2*)buf, BYTES / 8);
        hipLaunchKernelGGL(write_x4_nt, dim3(WGS), dim3(THREADS), 0, 0, (u32x4*)buf, BYTES / 16);
        CK(hipDeviceSynchronize());
    }
    printf("bytes per launch: read_x4 %zu read_x2 %zu read_rows_x2 %zu read_rows_x4 %zu dma_linear %zu dma_rows %zu write_x4 %zu write_x2 %zu write_x4_nt %zu\n",
           BYTES, BYTES, BYTES / 1536 / 128 * 128 * 1536, BYTES / 3072 / 128 * 128 * 3072, BYTES / 1024 * 1024,
           BYTES / 1536 / 32 * 32 * 1536, BYTES, BYTES, BYTES);
    return 0;
}
