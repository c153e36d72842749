// tools/coresidency_real.hip -- the REAL k_parse_wg<4, 1> beside resident emit-shaped wavefronts (tools/coresidency_probe.hip's A):
// is it placed beside them, and what does it cost?  (profiles/r05_wg_beside.txt: in the library's own sequence the parse did not start
// for 1.8 s.)   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/coresidency_real.hip -o /tmp/cr && /tmp/cr [blocks=600]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/hipdeflate.h"
#include "../7bgzf_amd/csrc/hd_deflate_static.hpp"
#include "../7bgzf_amd/csrc/hd_deflate_dynamic.hpp"
#include "../7bgzf_amd/csrc/hd_deflate_wg.hpp"
#include <chrono>
template <int V>
__global__ __launch_bounds__(192) void ka(unsigned *arrived, volatile unsigned *release)
{
	extern __shared__ unsigned lds[];
	if (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
	if (V == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
	if (V == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
	lds[threadIdx.x] = threadIdx.x;
	if ((threadIdx.x & 63) == 0)
		__hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	const long long t0 = wall_clock64();
	while (*release == 0 && wall_clock64() - t0 < 100000000ll / 5)      // <= 0.2 s
		__builtin_amdgcn_s_sleep(127);
	if (lds[threadIdx.x] == 12345678u) arrived[1] = 0;
}
int main(int argc, char **argv)
{
	const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 600, BB = 65280;
	uint8_t *h = (uint8_t *)malloc((size_t)N * BB);
	uint32_t x = 12345;
	for (size_t i = 0; i < (size_t)N * BB; i++) { x = x * 1103515245u + 12345u; h[i] = "ACGT\nFFFF:,I#"[(x >> 16) % 13]; }
	uint8_t *d_in; hipMalloc(&d_in, (size_t)N * BB); hipMemcpy(d_in, h, (size_t)N * BB, hipMemcpyHostToDevice);
	uint64_t *h_off = (uint64_t *)malloc(N * 8); uint32_t *h_len = (uint32_t *)malloc(N * 4);
	for (uint32_t i = 0; i < N; i++) { h_off[i] = (uint64_t)i * BB; h_len[i] = BB; }
	uint64_t *d_off; uint32_t *d_len; hipMalloc(&d_off, N * 8); hipMalloc(&d_len, N * 4);
	hipMemcpy(d_off, h_off, N * 8, hipMemcpyHostToDevice); hipMemcpy(d_len, h_len, N * 4, hipMemcpyHostToDevice);
	const uint64_t scr = hd::wg_scratch_bytes(N, BB);
	uint8_t *d_scr; hipMalloc(&d_scr, scr); hipMemset(d_scr, 0, scr);
	hd::CrcTables *d_ct; hipMalloc(&d_ct, sizeof(hd::CrcTables)); hipMemset(d_ct, 0, sizeof(hd::CrcTables));
	uint32_t *d_stalls; hipMalloc(&d_stalls, 16); hipMemset(d_stalls, 0, 16);
	hd::DeflateArgs a;
	memset(&a, 0, sizeof(a));
	a.in = d_in; a.in_off = d_off; a.in_len = d_len; a.nblocks = N; a.frame = HD_FRAME_RAW; a.level = 6;
	a.ct = d_ct; a.split_max = BB; a.stalls = d_stalls;
	a.split_ovf = (uint32_t *)d_scr;
	a.scratch = d_scr + (((uint64_t)N * 4 + 15) & ~(uint64_t)15);
	a.wg = 1; a.first = 0; a.count = N; a.wg_split = 1;
	unsigned *arrived, *release;
	hipHostMalloc(&arrived, 64); hipHostMalloc(&release, 4);
	hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int a_lds = argc > 2 ? atoi(argv[2]) : 10240, a_vgpr = argc > 3 ? atoi(argv[3]) : 128, a_per = argc > 4 ? atoi(argv[4]) : 3;
	printf("A: %d wavefronts per CU, %d B of LDS and %d VGPRs each\n", a_per, a_lds, a_vgpr);
	for (int mode = 0; mode < 3; mode++) {       // 0: the parse alone; 1: beside A as one-wavefront workgroups; 2: alone again
		const int wpw = 1, na = mode == 1 ? 256 * a_per : 0;
		arrived[0] = 0; *release = 0;
		if (na) {
			if (a_vgpr == 128) hipLaunchKernelGGL(ka<128>, dim3(na / wpw), dim3(64 * wpw), a_lds * wpw, s1, arrived, release);
			else if (a_vgpr == 96) hipLaunchKernelGGL(ka<96>, dim3(na / wpw), dim3(64 * wpw), a_lds * wpw, s1, arrived, release);
			else hipLaunchKernelGGL(ka<64>, dim3(na / wpw), dim3(64 * wpw), a_lds * wpw, s1, arrived, release);
			auto t0 = std::chrono::steady_clock::now();
			while (*(volatile unsigned *)arrived < (unsigned)na && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50))
				;
		}
		hipEventRecord(e0, s2);
		hipLaunchKernelGGL((hd::k_parse_wg<4, 1>), dim3(N), dim3(1024), 0, s2, a);
		hipEventRecord(e1, s2);
		hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		*release = 1;
		hipStreamSynchronize(s1);
		uint32_t st = 0; hipMemcpy(&st, d_stalls, 4, hipMemcpyDeviceToHost);
		printf("%s: k_parse_wg<4,1> over %u blocks of %u bytes in %.3f ms (A resident: %u wavefronts; stalls so far %u)\n",
		       mode == 1 ? "beside A" : "alone", N, BB, ms, na ? arrived[0] : 0, st);
	}
	return 0;
}
