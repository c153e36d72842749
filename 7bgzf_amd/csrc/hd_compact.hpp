// hd_compact.hpp -- size prefix scan + member gather.
//
// Role: the in-order writer of applet/7bgzf.c:223-272 (join threads in order,
// fwrite each member) becomes "exclusive prefix sum of the member sizes, then
// every member copied to its final offset", all on the device.  With several
// GPUs each rank adds its base (the all-gathered totals of the lower ranks,
// SURVEY.md 8(e)) through the `base` argument.  HBM-bound byte copy.
#pragma once
#include "hd_device.hpp"

namespace hd {

constexpr uint32_t SCAN_TILE = 2048;     // elements per 256-thread workgroup

// pass 1: per-tile sums
__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t *len, uint32_t n, uint64_t *tile_sum)
{
	__shared__ uint64_t wsum[4];
	const uint32_t t = threadIdx.x, tile = blockIdx.x;
	uint64_t s = 0;
	for (uint32_t i = tile * SCAN_TILE + t; i < n && i < (tile + 1) * SCAN_TILE; i += 256)
		s += len[i];
	for (int o = 32; o > 0; o >>= 1) {
		s += ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)s, o, 64)) |
		     ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(s >> 32), o, 64) << 32);
	}
	if ((t & 63) == 0)
		wsum[t >> 6] = s;
	__syncthreads();
	if (t == 0)
		tile_sum[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: one workgroup turns tile sums into exclusive tile offsets (+ base)
__global__ __launch_bounds__(256) void k_scan_tiles(uint64_t *tile_sum, uint32_t ntiles, uint64_t base, uint64_t *total)
{
	__shared__ uint64_t part[256];
	const uint32_t t = threadIdx.x;
	const uint32_t per = (ntiles + 255) / 256;
	uint64_t s = 0;
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t i = t * per + k;
		if (i < ntiles)
			s += tile_sum[i];
	}
	part[t] = s;
	__syncthreads();
	if (t == 0) {
		uint64_t run = base;
		for (uint32_t k = 0; k < 256; k++) {
			const uint64_t v = part[k];
			part[k] = run;
			run += v;
		}
		if (total)
			*total = run - base;
	}
	__syncthreads();
	uint64_t run = part[t];
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t i = t * per + k;
		if (i < ntiles) {
			const uint64_t v = tile_sum[i];
			tile_sum[i] = run;
			run += v;
		}
	}
}

// pass 3: exclusive scan inside each tile.  The intra-tile sums are 64 bits wide like the other two passes:
// a tile of 2048 members of >= 2 MiB each (MiGz -b, the uint32 in_len API) passes 4 GiB.
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t x)
{
	// x < 2^35 per lane (eight u32): split at bit 26, so that both 32-bit DPP scans stay carry-free
	// (64 x 2^26 = 2^32 is never reached, 64 x 2^9 is tiny)
	const uint32_t a = wave_incl_scan((uint32_t)x & 0x3ffffffu);
	const uint64_t b = wave_incl_scan((uint32_t)(x >> 26));
	return (uint64_t)a + (b << 26);
}

__global__ __launch_bounds__(256) void k_scan_finish(const uint32_t *len, uint32_t n, const uint64_t *tile_off, uint64_t *dst_off)
{
	__shared__ uint64_t wtot[4];
	const uint32_t t = threadIdx.x, tile = blockIdx.x, lane = t & 63, w = t >> 6;
	// thread t owns 8 consecutive elements
	const uint32_t i0 = tile * SCAN_TILE + t * 8;
	uint32_t v[8];
	uint64_t s = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		v[k] = i0 + k < n ? len[i0 + k] : 0;
		s += v[k];
	}
	const uint64_t incl = wave_incl_scan64(s);
	if (lane == 63)
		wtot[w] = incl;
	__syncthreads();
	uint64_t run = tile_off[tile] + (incl - s);
	for (uint32_t k = 0; k < w; k++)
		run += wtot[k];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		if (i0 + k < n)
			dst_off[i0 + k] = run;
		run += v[k];
	}
}

// one wavefront copies L bytes from s (16-byte aligned) to d (any alignment), whole destination dwords in the middle
__device__ __forceinline__ void compact_one(const uint8_t *s, uint32_t L, uint8_t *d, uint32_t lane)
{
	// head: bytes until d is dword aligned
	uint32_t head = (uint32_t)((4 - ((uintptr_t)d & 3)) & 3);
	if (head > L)
		head = L;
	if (lane < head)
		d[lane] = s[lane];
	const uint32_t body = (L - head) >> 2;           // whole destination dwords
	const uint32_t *s32 = (const uint32_t *)s;       // source dwords (aligned)
	uint32_t *d32 = (uint32_t *)(d + head);
	for (uint32_t k = lane; k < body; k += 64) {
		// destination dword k = source bytes [head + 4k, head + 4k + 4)
		const uint32_t so = head + 4 * k;
		const uint32_t lo = s32[so >> 2];
		const uint32_t hi = (so & 3) ? s32[(so >> 2) + 1] : 0;
		d32[k] = __builtin_amdgcn_alignbyte(hi, lo, so & 3);
	}
	const uint32_t done = head + 4 * body;
	if (lane < L - done)
		d[done + lane] = s[done + lane];
}

// one wavefront per member: slots + i*stride (16-byte aligned) -> dst + dst_off[i]
__global__ __launch_bounds__(64) void k_compact(const uint8_t *slots, uint64_t stride, const uint32_t *len,
						const uint64_t *dst_off, uint32_t n, uint8_t *dst)
{
	const uint32_t i = blockIdx.x, lane = threadIdx.x;
	if (i >= n)
		return;
	compact_one(slots + (uint64_t)i * stride, len[i], dst + dst_off[i], lane);
}

// RFC 1950 members (HD_FRAME_ZLIB): the Adler-32 of every block's input, written big-endian
// into the last four bytes of its member (lib/zlibutil.c:393-396 computes it on the CPU after
// the codec returns).  Kept out of the encode kernels, which sit at their register budget;
// one extra streaming read of the input, only in this frame.  One wavefront per block.
__global__ __launch_bounds__(64) void k_adler32_patch(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len,
						      uint32_t nblocks, uint8_t *out, uint64_t out_stride,
						      const uint32_t *out_len, const int32_t *status)
{
	const uint32_t b = blockIdx.x, lane = threadIdx.x;
	if (b >= nblocks || (status && status[b]) || out_len[b] < 6)
		return;
	const uint8_t *src = in + in_off[b];
	const uint32_t n = in_len[b];
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	AdlerLanes adl;
	adl.init();
	for (uint32_t piece = 0; piece * HD_PIECE < n; piece++) {
		const uint32_t o = piece * HD_PIECE + lane * 16;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (aligned && o + 16 <= n) {
			v = *(const uint4 *)(src + o);
		} else if (o < n) {
			uint32_t w[4] = { 0, 0, 0, 0 };
#pragma unroll
			for (uint32_t k = 0; k < 16; k++)
				w[k >> 2] |= (o + k < n ? (uint32_t)src[o + k] : 0u) << (8 * (k & 3));
			v = make_uint4(w[0], w[1], w[2], w[3]);
		}
		adl.fold(piece, lane, v);
	}
	const uint32_t a = adl.finish(n);
	if (lane < 4)
		out[(uint64_t)b * out_stride + out_len[b] - 4 + lane] = (uint8_t)(a >> (24 - 8 * lane));
}

} // namespace hd
