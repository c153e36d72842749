// hd_selftest.hip -- device self-test of the wave primitives every kernel leans
// on (DPP prefix scan, strided CRC folding, slot arithmetic).  Run by
// hipdeflate_selftest(); tests/test_gpu_selftest.py calls it on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/hipdeflate.h"
#include "hd_device.hpp"
#include "hd_deflate_dynamic.hpp"

namespace {

__global__ __launch_bounds__(64) void k_selftest_scan(const uint32_t *x, uint32_t *incl)
{
	incl[blockIdx.x * 64 + threadIdx.x] = hd::wave_incl_scan(x[blockIdx.x * 64 + threadIdx.x]);
}

__global__ __launch_bounds__(64) void k_selftest_slots(uint32_t *out)
{
	// out[len]   = packed (sym, eb, ev) for len 3..258; out[512 + i] for a sweep of offsets
	for (uint32_t len = 3 + threadIdx.x; len <= 258; len += 64) {
		uint32_t s, eb, ev;
		hd::len_slot(len, s, eb, ev);
		out[len] = (s << 16) | (eb << 8) | ev;
	}
	for (uint32_t i = threadIdx.x; i < 32768; i += 64) {
		uint32_t s, eb, ev;
		hd::off_slot(i + 1, s, eb, ev);
		out[512 + i] = (s << 24) | (eb << 16) | ev;
	}
}

// The parse publishes 64 positions per step with ONE ds_write_b16; lanes whose four bytes hash alike write the
// same table entry in that instruction.  The encoder (and its CPU twin, which keeps the largest position)
// relies on what gfx950's LDS does then: the highest lane's data stays.  This kernel checks exactly that,
// for 16-bit entries, with conflict patterns from "all lanes on one entry" to "random over 1536 entries".
// (the second half of the blocks does the same with ds_write_b32 on dword entries: the two-way buckets of the lazy levels)
__global__ __launch_bounds__(64) void k_selftest_lds_order(const uint16_t *idx, uint32_t rounds, uint16_t *out)
{
	__shared__ uint16_t table[1536];
	__shared__ uint32_t table32[1536];
	const uint32_t lane = threadIdx.x;
	const bool wide = blockIdx.x >= gridDim.x / 2;
	for (uint32_t r = 0; r < rounds; r++) {
		for (uint32_t i = lane; i < 1536; i += 64) {
			table[i] = 0;
			table32[i] = 0;
		}
		__syncthreads();
		const uint16_t h = idx[(blockIdx.x * rounds + r) * 64 + lane];
		if (wide)
			table32[h] = 0xabcd0000u | (lane + 1);
		else
			table[h] = (uint16_t)(lane + 1);
		__syncthreads();
		out[(blockIdx.x * rounds + r) * 64 + lane] = wide ? (uint16_t)table32[h] : table[h];
	}
}

// one wavefront per frequency vector through the encoder's Huffman construction
__global__ __launch_bounds__(64) void k_selftest_build(const uint32_t *freq, uint32_t nsyms, uint32_t maxbits, uint8_t *lens)
{
	__shared__ uint32_t f[288], o[288];
	__shared__ hd::HuffScratch hs;
	const uint32_t lane = threadIdx.x;
	for (uint32_t s = lane; s < nsyms; s += 64)
		f[s] = freq[(size_t)blockIdx.x * nsyms + s];
	hd::build_code(f, nsyms, maxbits, o, hs, lane);
	for (uint32_t s = lane; s < nsyms; s += 64)
		lens[(size_t)blockIdx.x * nsyms + s] = (uint8_t)(o[s] >> 16);
}

} // namespace

// ---- LDS write arbitration: of the lanes that store to one entry in one instruction, the highest stays ----
// Kernel == CPU twin (and with it golden hashes and multi-rank determinism) leans on this property of the gfx950 LDS,
// which no document promises: ctx_init() runs this probe once per process and refuses the device if it fails.
// Returns the number of entries that did not hold the highest writer; < 0 on an allocation error.
int hd_probe_lds_order(void)
{
	const uint32_t nb = 16, rounds = 32, n = nb * rounds * 64;
	uint16_t *hx = (uint16_t *)malloc(n * 2), *ho = (uint16_t *)malloc(n * 2), *dx = nullptr, *dout = nullptr;
	uint32_t seed = 777;
	for (uint32_t i = 0; i < n; i++) {
		seed = seed * 1664525u + 1013904223u;
		const uint32_t pat = (i / 64) % 8, l = i % 64;
		// 0: one entry; 1: pairs; 2: two entries alternating; 3: 8 distinct, same bank; 4..7: random over 2^k
		hx[i] = pat == 0 ? 5 : pat == 1 ? (uint16_t)(l / 2) : pat == 2 ? (uint16_t)(l & 1)
			: pat == 3 ? (uint16_t)((l & 7) * 64) : (uint16_t)((seed >> 16) % (pat == 4 ? 4u : pat == 5 ? 16u : pat == 6 ? 64u : 1536u));
	}
	if (hipMalloc((void **)&dx, n * 2) != hipSuccess || hipMalloc((void **)&dout, n * 2) != hipSuccess) {
		free(hx);
		free(ho);
		(void)hipFree(dx);
		return -1;
	}
	(void)hipMemcpy(dx, hx, n * 2, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_selftest_lds_order, dim3(nb), dim3(64), 0, 0, dx, rounds, dout);
	int bad = hipMemcpy(ho, dout, n * 2, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
	for (uint32_t g = 0; g < n && !(bad && g == 0); g += 64)
		for (uint32_t l = 0; l < 64; l++) {
			uint32_t top = l;
			for (uint32_t k = l + 1; k < 64; k++)
				if (hx[g + k] == hx[g + l])
					top = k;
			if (ho[g + l] != top + 1) {
				if (bad++ < 5)
					fprintf(stderr, "hipdeflate: LDS store arbitration: group %u lane %u entry %u holds lane %u, "
						"highest writer is %u\n", g / 64, l, hx[g + l], ho[g + l] - 1, top);
			}
		}
	free(hx);
	free(ho);
	(void)hipFree(dx);
	(void)hipFree(dout);
	return bad;
}

/* test entry: code lengths the DEVICE Huffman construction gives `nvec` frequency vectors of `nsyms` symbols
 * each (tests/test_gpu_parity.py compares them with the oracle's on adversarial distributions) */
extern "C" int hipdeflate_test_build_lengths(const uint32_t *freq, uint32_t nvec, uint32_t nsyms, uint32_t maxbits,
					     uint8_t *lens_out)
{
	int r = hipdeflate_available();
	if (r)
		return r;
	// no length-limited code exists for more than 2^maxbits symbols (the builder's push-up loop would run off
	// its length counters)
	if (!freq || !lens_out || !nvec || nsyms < 2 || nsyms > 288 || maxbits < 1 || maxbits > 15 || nsyms > (1u << maxbits))
		return HD_E_ARG;
	uint32_t *df = nullptr;
	uint8_t *dl = nullptr;
	const size_t n = (size_t)nvec * nsyms;
	if (hipMalloc((void **)&df, n * 4) != hipSuccess)
		return HD_E_NOMEM;
	if (hipMalloc((void **)&dl, n) != hipSuccess) {
		(void)hipFree(df);
		return HD_E_NOMEM;
	}
	(void)hipMemcpy(df, freq, n * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_selftest_build, dim3(nvec), dim3(64), 0, 0, df, nsyms, maxbits, dl);
	const hipError_t e = hipMemcpy(lens_out, dl, n, hipMemcpyDeviceToHost);
	(void)hipFree(df);
	(void)hipFree(dl);
	return e == hipSuccess ? 0 : HD_E_NODEVICE;
}

extern "C" int hipdeflate_selftest(void)
{
	int r = hipdeflate_available();
	if (r)
		return r;
	int fails = 0;
	// ---- scan -------------------------------------------------------------
	{
		const int nb = 8;
		uint32_t hx[nb * 64], hi[nb * 64];
		uint32_t seed = 12345;
		for (int i = 0; i < nb * 64; i++) {
			seed = seed * 1664525u + 1013904223u;
			hx[i] = i < 64 ? 31 : i < 128 ? (i & 1) : (seed >> 27);
		}
		uint32_t *dx, *di;
		if (hipMalloc((void **)&dx, sizeof(hx)) != hipSuccess || hipMalloc((void **)&di, sizeof(hi)) != hipSuccess)
			return HD_E_NOMEM;
		(void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_selftest_scan, dim3(nb), dim3(64), 0, 0, dx, di);
		(void)hipMemcpy(hi, di, sizeof(hi), hipMemcpyDeviceToHost);
		for (int b = 0; b < nb; b++) {
			uint32_t run = 0;
			for (int l = 0; l < 64; l++) {
				run += hx[b * 64 + l];
				if (hi[b * 64 + l] != run) {
					if (fails < 5)
						fprintf(stderr, "hipdeflate selftest: scan[%d][%d] = %u, want %u\n", b, l,
							hi[b * 64 + l], run);
					fails++;
				}
			}
		}
		(void)hipFree(dx);
		(void)hipFree(di);
	}
	// ---- LDS write arbitration (also run by ctx_init) ----------------------------
	{
		const int bad = hd_probe_lds_order();
		if (bad < 0)
			return HD_E_NOMEM;
		fails += bad;
	}
	// ---- slot arithmetic vs RFC 1951 3.2.5 tables ---------------------------
	{
		static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
						    67, 83, 99, 115, 131, 163, 195, 227, 258 };
		static const uint8_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
						  4, 4, 4, 4, 5, 5, 5, 5, 0 };
		static const uint16_t obase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513,
						    769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
		static const uint8_t oext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
						  9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
		const size_t n = 512 + 32768;
		uint32_t *d, *h = (uint32_t *)malloc(n * 4);
		if (hipMalloc((void **)&d, n * 4) != hipSuccess)
			return HD_E_NOMEM;
		hipLaunchKernelGGL(k_selftest_slots, dim3(1), dim3(64), 0, 0, d);
		(void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
		for (uint32_t len = 3; len <= 258; len++) {
			uint32_t s = h[len] >> 16, eb = (h[len] >> 8) & 0xff, ev = h[len] & 0xff;
			// 258 is symbol 285 (index 28), never 284 + extra 31
			if (s > 28 || lext[s] != eb || lbase[s] + ev != len || ev >= (1u << eb) + (eb == 0) ||
			    (len == 258 && s != 28))
				fails++;
		}
		for (uint32_t i = 0; i < 32768; i++) {
			uint32_t s = h[512 + i] >> 24, eb = (h[512 + i] >> 16) & 0xff, ev = h[512 + i] & 0xffff;
			if (s > 29 || oext[s] != eb || obase[s] + ev != i + 1 || (eb && ev >= (1u << eb)))
				fails++;
		}
		if (fails)
			fprintf(stderr, "hipdeflate selftest: slot arithmetic failures so far: %d\n", fails);
		free(h);
		(void)hipFree(d);
	}
	// ---- CRC folding: through the public inflate/deflate entry points -------
	{
		// stored level-0 members of assorted sizes carry the kernel's CRC-32;
		// compare with a bitwise CRC computed here
		const uint32_t sizes[] = { 0, 1, 15, 16, 17, 1023, 1024, 1025, 4096 + 7, 65280, 65536, 70001 };
		for (uint32_t n : sizes) {
			uint8_t *src = (uint8_t *)malloc(n + 1), *dst = (uint8_t *)malloc(n + 1024);
			uint32_t seed = n * 2654435761u + 1;
			for (uint32_t i = 0; i < n; i++) {
				seed = seed * 1664525u + 1013904223u;
				src[i] = seed >> 24;
			}
			uint32_t c = 0xffffffffu;
			for (uint32_t i = 0; i < n; i++) {
				c ^= src[i];
				for (int k = 0; k < 8; k++)
					c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
			}
			c = ~c;
			uint64_t off = 0;
			uint32_t len = n, olen = 0, crc = 0;
			int32_t st = -1;
			int rr = hipdeflate_batch_deflate(src, &off, &len, 1, 0, HD_FRAME_RAW, dst, (n + 1024) & ~15u, n + 1000,
							  &olen, &crc, &st);
			if (rr || st || crc != c) {
				fprintf(stderr, "hipdeflate selftest: crc n=%u got %08x want %08x (rc %d st %d)\n", n, crc, c,
					rr, st);
				fails++;
			}
			free(src);
			free(dst);
		}
	}
	return fails ? 1 : 0;
}
