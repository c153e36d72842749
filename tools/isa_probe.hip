// tools/isa_probe.hip -- small questions to the gfx950 ISA that the guides leave open (round 4):
//   1. does v_alignbyte_b32 use only bits [1:0] of its shift operand?  (then a ring offset can be passed as it is)
//   2. v_add_u32 ... clamp saturates at 0xffffffff?
//   3. v_ffbl_b32 of 0 is 0xffffffff?
// hipcc --offload-arch=gfx950 -O2 tools/isa_probe.hip -o /tmp/isa_probe && /tmp/isa_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out)
{
	const uint32_t l = threadIdx.x;
	uint32_t a, b, c;
	asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(a) : "v"(0x44332211u), "v"(0xddccbbaau), "v"(l));
	asm volatile("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(b) : "v"(0xfffffff0u + l), "v"(8u));
	asm volatile("v_ffbl_b32 %0, %1" : "=v"(c) : "v"(l == 0 ? 0u : 1u << (l & 31)));
	out[l] = a;
	out[64 + l] = b;
	out[128 + l] = c;
}
int main()
{
	uint32_t *d, h[192];
	if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
	int low2 = 1;
	for (int l = 0; l < 64; l++) {
		const uint64_t v = 0x44332211ddccbbaaull;
		const uint32_t want = (uint32_t)(v >> (8 * (l & 3)));
		if (h[l] != want) low2 = 0;
	}
	printf("alignbyte uses shift[1:0] only: %s (shift 4 -> %08x, shift 5 -> %08x)\n", low2 ? "yes" : "NO", h[4], h[5]);
	printf("add clamp: %08x %08x %08x (l = 6, 7, 8: fffffffe ffffffff ffffffff expected)\n", h[64 + 6], h[64 + 7], h[64 + 8]);
	printf("ffbl(0) = %08x, ffbl(1<<5) = %u\n", h[128], h[128 + 5]);
	return 0;
}
