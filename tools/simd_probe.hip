// tools/simd_probe.hip -- which SIMD does wavefront w of a 1024-thread workgroup run on?  (k_emit_wg gives the serial code
// constructions of a member's DEFLATE blocks to wavefronts that must not share a SIMD.)  hipcc --offload-arch=gfx950 -O2
// tools/simd_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void k(unsigned *out)
{
	unsigned id;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
	if ((threadIdx.x & 63) == 0)
		out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
	unsigned *d, h[64];
	hipMalloc(&d, sizeof(h));
	hipLaunchKernelGGL(k, dim3(4), dim3(1024), 0, 0, d);
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	for (int b = 0; b < 4; b++) {
		printf("{\"workgroup\": %d, \"simd_of_wave\": [", b);
		for (int w = 0; w < 16; w++)
			printf("%u%s", (h[b * 16 + w] >> 4) & 3, w < 15 ? ", " : "");
		printf("], \"wave_slot\": [");
		for (int w = 0; w < 16; w++)
			printf("%u%s", h[b * 16 + w] & 15, w < 15 ? ", " : "");
		printf("], \"cu\": %u}\n", (h[b * 16] >> 8) & 15);
	}
	return 0;
}
