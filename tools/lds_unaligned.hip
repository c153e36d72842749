// tools/lds_unaligned.hip -- does the gfx950 LDS take ds_read_b64 / ds_write_b32 / ds_write_b64 at ANY byte address (unaligned
// access mode), with the right bytes, and at what rate?  (The inflate kernel's short match copies want one 8-byte read and
// one 8-byte write per match at arbitrary ring positions.)  hipcc --offload-arch=gfx950 -O2 tools/lds_unaligned.hip -o /tmp/lu && /tmp/lu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_check(const uint32_t* offs, uint8_t* out_rd, uint8_t* out_img, uint32_t* sizes)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds;
    const uint32_t o = offs[blockIdx.x * 64 + lane];           // any byte offset < 2040
    // reads
    uint64_t r64; uint32_t r32; uint32_t r16;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r64) : "v"(base + o) : "memory");
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r32) : "v"(base + o) : "memory");
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r16) : "v"(base + o) : "memory");
    uint8_t* rd = out_rd + (size_t)(blockIdx.x * 64 + lane) * 16;
    memcpy(rd, &r64, 8); memcpy(rd + 8, &r32, 4); memcpy(rd + 12, &r16, 2);
    __syncthreads();
    // writes into the upper half: lane l writes 8 bytes at 2048 + 24 l + (o & 7), 4 bytes 12 further, 2 bytes 18 further
    const uint32_t w = 2048 + 24 * lane + (o & 7);
    const uint64_t v64 = 0x0807060504030201ull + 0x1010101010101010ull * (lane & 15);
    const uint32_t v32 = 0xa4a3a2a1u + lane, v16 = 0xb2b1u + (lane << 8);
    asm volatile("ds_write_b64 %0, %1" ::"v"(base + w), "v"(v64) : "memory");
    asm volatile("ds_write_b32 %0, %1" ::"v"(base + w + 12), "v"(v32) : "memory");
    asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(base + w + 18), "v"(v16) : "memory");
    __syncthreads();
    for (uint32_t i = lane; i < 2048; i += 64) out_img[(size_t)blockIdx.x * 2048 + i] = lds[2048 + i];
    if (lane == 0) sizes[blockIdx.x] = 1;
}

template <int MODE>
__global__ void k_rate(uint64_t* stamps, uint32_t misalign, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds;
    const uint32_t a = base + lane * 8 + misalign;
    uint64_t x = lane;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 0) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(a), "n"(k * 512 % 2048) : "memory");
            if (MODE == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(x), "n"(k * 512 % 2048) : "memory");
            if (MODE == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a), "v"((uint32_t)x), "n"(k * 512 % 2048) : "memory");
            if (MODE == 3) { uint32_t x32; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x32) : "v"(a), "n"(k * 512 % 2048) : "memory"); x = x32; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) stamps[blockIdx.x] = t1 - t0;
    if (x == 0x123456789ull) stamps[0] = 0;
}

int main()
{
    const int nb = 64;
    std::vector<uint32_t> offs(nb * 64);
    uint32_t seed = 99;
    for (auto& o : offs) { seed = seed * 1664525u + 1013904223u; o = (seed >> 8) % 2040; }
    for (int i = 0; i < 64; i++) offs[i] = i;                       // every alignment, contiguous
    uint32_t *d_offs, *d_sz; uint8_t *d_rd, *d_img;
    CK(hipMalloc(&d_offs, offs.size() * 4)); CK(hipMalloc(&d_rd, offs.size() * 16)); CK(hipMalloc(&d_img, nb * 2048)); CK(hipMalloc(&d_sz, nb * 4));
    CK(hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(nb), dim3(64), 0, 0, d_offs, d_rd, d_img, d_sz);
    std::vector<uint8_t> rd(offs.size() * 16), img(nb * 2048);
    CK(hipMemcpy(rd.data(), d_rd, rd.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(img.data(), d_img, img.size(), hipMemcpyDeviceToHost));
    int bad_rd = 0, bad_wr = 0;
    for (size_t i = 0; i < offs.size(); i++) {
        for (int k = 0; k < 8; k++) bad_rd += rd[i * 16 + k] != (uint8_t)((offs[i] + k) * 7 + 3);
        for (int k = 0; k < 4; k++) bad_rd += rd[i * 16 + 8 + k] != (uint8_t)((offs[i] + k) * 7 + 3);
        for (int k = 0; k < 2; k++) bad_rd += rd[i * 16 + 12 + k] != (uint8_t)((offs[i] + k) * 7 + 3);
    }
    for (int b = 0; b < nb; b++) {
        std::vector<uint8_t> want(2048);
        for (int i = 0; i < 2048; i++) want[i] = (uint8_t)((2048 + i) * 7 + 3);
        for (int l = 0; l < 64; l++) {
            const uint32_t w = 24 * l + (offs[b * 64 + l] & 7);
            const uint64_t v64 = 0x0807060504030201ull + 0x1010101010101010ull * (l & 15);
            const uint32_t v32 = 0xa4a3a2a1u + l, v16 = 0xb2b1u + (l << 8);
            if (w + 20 > 2048) continue;
            memcpy(&want[w], &v64, 8); memcpy(&want[w + 12], &v32, 4); memcpy(&want[w + 18], &v16, 2);
        }
        for (int i = 0; i < 1536; i++) bad_wr += want[i] != img[b * 2048 + i];
    }
    printf("{\"unaligned_lds\": {\"read_mismatches\": %d, \"write_mismatches\": %d", bad_rd, bad_wr);
    // rates: 4 waves per SIMD, aligned vs misaligned by 1 and by 4
    uint64_t* d_st; CK(hipMalloc(&d_st, 8 * 4096));
    std::vector<uint64_t> st(4096);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * 16, iters = 2000;
    const char* names[4] = {"ds_read_b64", "ds_write_b64", "ds_write_b32", "ds_read_b32"};
    for (int mode = 0; mode < 4; mode++)
        for (uint32_t mis : {0u, 1u, 4u}) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(grid), dim3(64), 0, 0, d_st, mis, iters);
                if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(grid), dim3(64), 0, 0, d_st, mis, iters);
                if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(grid), dim3(64), 0, 0, d_st, mis, iters);
                if (mode == 3) hipLaunchKernelGGL(k_rate<3>, dim3(grid), dim3(64), 0, 0, d_st, mis, iters);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(st.data(), d_st, 8 * grid, hipMemcpyDeviceToHost));
            double s = 0; for (int i = 0; i < grid; i++) s += (double)st[i];
            printf(", \"%s_mis%u_cyc_per_inst_simd_at_4_waves\": %.2f", names[mode], mis, s / grid / (iters * 16.0) / 4.0);
        }
    printf("}}\n");
    return bad_rd || bad_wr;
}
