// tools/beside_filter_probe.hip -- what HW_REG_LDS_ALLOC says about a wavefront's LDS block: the resident emit wavefronts of the "beside" scheme
// (hd_deflate_wg.hpp) must be the ones in the LOWEST LDS blocks of their CU, or the candidates that leave fragment what the parse needs.
// hipcc --offload-arch=gfx950 -O2 tools/beside_filter_probe.hip -o /tmp/fp && /tmp/fp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <set>
__global__ __launch_bounds__(64) void k(unsigned *out)
{
	__shared__ unsigned lds[2416];            // 9,664 B: the emit-only kernel's
	lds[threadIdx.x] = 1;
	if (threadIdx.x == 0) {
		unsigned hw, xcc, la;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
		out[2 * blockIdx.x] = 0x80000000u | ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u) | (((hw >> 4) & 3u) << 16);
		out[2 * blockIdx.x + 1] = la;
	}
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < 100000) __builtin_amdgcn_s_sleep(64);       // 1 ms: everybody is resident at once
	if (lds[threadIdx.x] == 7) out[0] = 0;
}
int main()
{
	const int N = 1536;
	unsigned *out; hipMalloc(&out, N * 8); hipMemset(out, 0, N * 8);
	hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, 0, out);
	hipDeviceSynchronize();
	static unsigned h[2 * N]; hipMemcpy(h, out, N * 8, hipMemcpyDeviceToHost);
	std::map<unsigned, std::set<unsigned>> per;
	for (int i = 0; i < N; i++) per[h[2 * i] & 0xfff].insert(h[2 * i + 1]);
	int shown = 0;
	for (auto &p : per) {
		if (shown++ >= 3) break;
		printf("cu %03x: raw LDS_ALLOC of its wavefronts:", p.first);
		for (unsigned v : p.second) printf(" %08x", v);
		printf("\n");
	}
	return 0;
}
