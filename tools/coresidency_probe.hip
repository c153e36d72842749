// tools/coresidency_probe.hip -- can a CU carry ONE workgroup of the parse's shape (1024 threads, 131 KB of LDS) AND three
// one-wavefront workgroups of the emit-only kernel's shape (10 KB each) at the same time, everywhere on the chip?  (Round 5: the
// emit kernel on a second stream beside the parse lost 7 % because emit wavefronts that come and go fragment the CUs' LDS;
// PERSISTENT emit wavefronts, placed first, would not -- if the dispatcher spreads 768 of them three to a CU.)
// A: 768 x 64 threads, 10 KB LDS, resident until released.  B: 512 x 1024 threads, 131200 B LDS, ~200 us each, launched on a
// second stream once A is all there.  Printed: A's wavefronts per CU, B's workgroups per CU, B's wall time (2 rounds = ~0.4 ms
// if every CU takes one).    hipcc --offload-arch=gfx950 -O2 tools/coresidency_probe.hip -o /tmp/cp && /tmp/cp [A_per_cu=3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <chrono>
__device__ __forceinline__ unsigned where()
{
	unsigned id, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	// se_id [15:13], sh_id [12], cu_id [11:8]; xcc_id [3:0]
	return ((xcc & 15) << 8) | (((id >> 13) & 7) << 5) | (((id >> 12) & 1) << 4) | ((id >> 8) & 15);
}
// (the real kernels' registers count too: the emit-only kernel holds 121 -> 128 VGPRs, a parse wavefront 95 -> 96 and a SIMD has 512:
// four parse wavefronts and ONE emit wavefront per SIMD fit, two emit wavefronts on one SIMD keep the parse workgroup off the CU)
__device__ unsigned g_sgpr_pressure;          // != 0: both kernels hold ~100 scalar registers too, as the real pair does
__global__ __launch_bounds__(192) void ka(unsigned *out, unsigned *arrived, volatile unsigned *release)
{
	extern __shared__ unsigned lds[];
	asm volatile("v_mov_b32 v127, 0" ::: "v127");
	if (g_sgpr_pressure)
		asm volatile("s_mov_b32 s100, 0" ::: "s100");
	lds[threadIdx.x] = threadIdx.x;
	if ((threadIdx.x & 63) == 0) {
		out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = where() | (((unsigned)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) & 3) << 16);
		__hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	const long long t0 = wall_clock64();
	while (*release == 0 && wall_clock64() - t0 < 100000000ll / 10)      // <= 0.1 s at 100 MHz
		__builtin_amdgcn_s_sleep(32);
	if (lds[threadIdx.x] == 12345678u) out[0] = 0;
}
__global__ __launch_bounds__(1024) void kb(unsigned *out, long long *t)
{
	extern __shared__ unsigned lds[];
	asm volatile("v_mov_b32 v95, 0" ::: "v95");
	if (g_sgpr_pressure)
		asm volatile("s_mov_b32 s95, 0" ::: "s95");
	lds[threadIdx.x] = threadIdx.x;
	const long long t0 = wall_clock64();
	if (threadIdx.x == 0)
		out[blockIdx.x] = where();
	while (wall_clock64() - t0 < 20000)                                   // 200 us at 100 MHz
		__builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64(); }
	if (lds[threadIdx.x] == 12345678u) out[0] = 0;
}
// the gate of the real scheme: one wavefront on B's stream that waits for A's wavefronts on the DEVICE, B right behind it
__global__ __launch_bounds__(64) void kgate(unsigned *arrived, unsigned want)
{
	for (unsigned spins = 0; spins < (1u << 16); spins++) {
		if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= want)
			break;
		__builtin_amdgcn_s_sleep(16);
	}
}
int main(int argc, char **argv)
{
	const int per = argc > 1 ? atoi(argv[1]) : 3, wpw = argc > 2 ? atoi(argv[2]) : 1;      // A: wavefronts per CU, wavefronts per workgroup
	const int NA = 256 * per, NB = 512;
	const int sg = argc > 3 ? atoi(argv[3]) : 0, gate = argc > 4 ? atoi(argv[4]) : 0;       // scalar-register pressure; B behind a device-side gate
	{ unsigned v = (unsigned)sg; hipMemcpyToSymbol(HIP_SYMBOL(g_sgpr_pressure), &v, 4); }
	unsigned *da, *db, *arrived, *release;
	long long *dt;
	hipMalloc(&da, NA * 4); hipMalloc(&db, NB * 4); hipMalloc(&dt, NB * 16);
	hipHostMalloc(&arrived, 4); hipHostMalloc(&release, 4);
	*arrived = 0; *release = 0;
	hipStream_t s1, s2;
	hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	hipFuncSetAttribute((const void *)kb, hipFuncAttributeMaxDynamicSharedMemorySize, 131200);
	hipLaunchKernelGGL(ka, dim3(NA / wpw), dim3(64 * wpw), 10240 * wpw, s1, da, arrived, release);
	auto t0 = std::chrono::steady_clock::now();
	if (gate) {
		hipLaunchKernelGGL(kgate, dim3(1), dim3(64), 0, s2, arrived, (unsigned)NA);
	} else {
		while (*(volatile unsigned *)arrived < (unsigned)NA && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50))
			;
		printf("A: %u of %d wavefronts resident when B is launched\n", *(volatile unsigned *)arrived, NA);
	}
	auto tb0 = std::chrono::steady_clock::now();
	hipLaunchKernelGGL(kb, dim3(NB), dim3(1024), 131200, s2, db, dt);
	hipStreamSynchronize(s2);
	auto tb1 = std::chrono::steady_clock::now();
	*release = 1;
	hipStreamSynchronize(s1);
	printf("B: %d workgroups of 200 us in %.0f us wall (two rounds on 256 CUs = ~410)\n", NB, std::chrono::duration<double, std::micro>(tb1 - tb0).count());
	unsigned *ha = (unsigned *)malloc(NA * 4), *hb = (unsigned *)malloc(NB * 4);
	hipMemcpy(ha, da, NA * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, NB * 4, hipMemcpyDeviceToHost);
	std::map<unsigned, int> ca, cb;
	std::map<unsigned, int> simd_pairs;       // CUs where two of A's wavefronts share a SIMD
	{
		std::map<unsigned, int> seen;
		for (int i = 0; i < NA; i++)
			if (seen[ha[i]]++)
				simd_pairs[ha[i] & 0xffff]++;
		for (int i = 0; i < NA; i++)
			ha[i] &= 0xffff;
	}
	printf("A: CUs where two of its wavefronts share a SIMD: %zu\n", simd_pairs.size());
	for (int i = 0; i < NA; i++) ca[ha[i]]++;
	for (int i = 0; i < NB; i++) cb[hb[i]]++;
	std::map<int, int> ha_hist, hb_hist;
	for (auto &p : ca) ha_hist[p.second]++;
	for (auto &p : cb) hb_hist[p.second]++;
	printf("A: %zu distinct CUs; CUs by wavefronts of A:", ca.size());
	for (auto &p : ha_hist) printf(" %d x %d", p.second, p.first);
	printf("\nB: %zu distinct CUs; CUs by workgroups of B:", cb.size());
	for (auto &p : hb_hist) printf(" %d x %d", p.second, p.first);
	int both = 0;
	for (auto &p : cb) both += ca.count(p.first);
	printf("\nCUs that carried B beside A: %d\n", both);
	return 0;
}
