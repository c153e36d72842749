// hd_segment.hpp -- levels >= 1, blocks longer than HD_SEG_LIMIT (a 1 MiB MiGz member, a large
// zlibutil buffer): the block is coded as independent HD_SEG_BYTES segments, every one in
// full-flush form, an empty final block behind the last (include/hipdeflate_params.h; the CPU
// twin's twin_segmented()).  LATENCY MODE (HD_FRAME_LATENCY) is the same machinery with segments
// of HD_LAT_SEG_BYTES(level) for every block longer than that: 16 (8) wavefronts per BGZF block
// when the batch is a handful of blocks from htslib's worker threads.  One member is then work for 17 wavefronts instead of 40 ms of a single
// one, and wavefronts that start together no longer read 1 MiB apart (one HBM channel for all).  Three small launches wrap the ordinary dynamic
// path: a segment table, the coding of the segments into scratch slots, and a stitch (sizes, header,
// trailer, CRC-32 of the whole from the CRCs of the parts) followed by the gather of the payloads.
#pragma once
#include "hd_compact.hpp"
#include "hd_deflate_dynamic.hpp"

namespace hd {

// a slot holds any segment's worst case
__host__ __device__ inline uint32_t seg_stride(uint32_t seg) { return (HD_STORED_SIZE(seg) + 5u + 32u + 15u) & ~15u; }
constexpr uint32_t SEG_ROUND_MAX = 65536;                                       // segments coded per round (4.3 GB of slots)

struct SegArgs {
	DeflateArgs a;               // the members: the caller's arrays
	uint32_t first, count;       // blocks of this round
	uint32_t S;                  // segment slots per block
	uint32_t seg, limit;         // segment bytes; blocks longer than `limit` are segmented
	uint64_t *seg_off;           // [count * S] each
	uint32_t *seg_len;
	uint32_t *seg_olen;
	uint32_t *seg_crc;
	int32_t *seg_st;
	uint64_t *seg_dst;
};

// segment slots per block: enough for the longest block whose worst case fits the slot
inline uint32_t seg_slots_per_block(uint32_t cap, uint32_t seg)
{
	const uint32_t full = HD_STORED_SIZE(seg) + 5u;
	// (a raw frame has no header, a flush frame no 03 00 tail -- its shortest last segment, one byte, takes
	// 1 + 5 + 5 = 11 bytes: the count may be one high for the other frames, never low)
	return cap / full + (cap % full >= 11u ? 1u : 0u);
}

inline uint32_t seg_round_blocks(uint32_t nblocks, uint32_t S)
{
	uint32_t r = SEG_ROUND_MAX / S;
	if (r == 0)
		r = 1;
	return r < nblocks ? r : nblocks;
}

// bytes per round: tables + slots (the coding's own scratch comes behind)
inline uint64_t seg_round_bytes(uint32_t round_blocks, uint32_t S, uint32_t seg)
{
	const uint64_t nseg = (uint64_t)round_blocks * S;
	return nseg * (8 + 4 + 4 + 4 + 4 + 8) + 64 + nseg * seg_stride(seg);
}

// level 1 needs the tables and slots only; the dynamic levels add the small blocks' and the segments' token scratch
inline uint64_t segmented_scratch_bytes(uint32_t nblocks, uint32_t cap, int level, uint32_t seg = HD_SEG_BYTES)
{
	const uint32_t S = seg_slots_per_block(cap, seg), rb = seg_round_blocks(nblocks, S);
	if (level < 2)
		return seg_round_bytes(rb, S, seg) + 64;
	return dynamic_scratch_bytes(nblocks, cap, level) + seg_round_bytes(rb, S, seg) +
	       dynamic_scratch_bytes(rb * S, seg_stride(seg), level, HD_LAT_PARTS(level, seg)) + 64;
}

__global__ __launch_bounds__(256) void k_seg_table(SegArgs g)
{
	const uint32_t t = blockIdx.x * 256 + threadIdx.x;
	if (t >= g.count * g.S)
		return;
	const uint32_t i = g.first + t / g.S, k = t % g.S;
	const uint32_t len = g.a.in_len[i];
	const uint64_t o = (uint64_t)k * g.seg;
	// blocks up to the limit are not segmented (the ordinary coding takes them): all their slots stay empty
	const uint32_t sl = (len > g.limit && o < len) ? (len - o < g.seg ? (uint32_t)(len - o) : g.seg) : 0u;
	g.seg_off[t] = g.a.in_off[i] + (sl ? o : 0);
	g.seg_len[t] = sl;
}

// One WAVEFRONT per member, lane k = segment k (members of more than 64 segments: in rounds): where each
// segment's payload goes (a prefix sum of the segment sizes), the container bytes around them, and the
// CRC-32 of the whole from the CRCs of the parts -- crc(A || B) = crc(A) * x^(8 |B|) ^ crc(B), so every lane
// moves its segment's CRC to the end of the member with ONE table step (SM: m full segments behind it) and
// the lanes are XOR-ed; a ragged last segment adds its length once, behind the sum.  (A single thread per
// member walking the segments with bit-serial GF(2) products took 33 us -- a third of a latency-mode call.)
__device__ __forceinline__ void seg_stitch_member(const SegArgs &g, uint32_t t, uint32_t lane)
{
	const DeflateArgs &a = g.a;
	const CrcTables *ct = a.ct;
	const uint32_t i = g.first + t, len = a.in_len[i];
	const uint64_t base = (uint64_t)t * g.S;
	if (len <= g.limit) {
		for (uint32_t k = lane; k < g.S; k += 64) {
			g.seg_olen[base + k] = 0;               // nothing of this block's to gather
			g.seg_dst[base + k] = 0;
		}
		return;
	}
	const uint32_t hdr = frame_hdr_bytes(a.frame), trl = frame_trl_bytes(a.frame);
	const bool flush = a.frame == HD_FRAME_RAW_FLUSH;
	uint64_t cap = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
	if (a.frame == HD_FRAME_BGZF && cap > 65536)
		cap = 65536;
	const uint32_t nseg = HD_SEGN_COUNT(len, g.seg);
	bool ok = nseg <= g.S && (uint64_t)hdr + HD_SEGN_WORST((uint64_t)len, g.seg, flush) + trl <= cap;
	if (ok)
		for (uint32_t k0 = 0; k0 < nseg; k0 += 64)
			ok = ok && !__ballot(k0 + lane < nseg && g.seg_st[base + k0 + lane] != 0);
	if (!ok) {
		for (uint32_t k = lane; k < g.S; k += 64) {
			g.seg_olen[base + k] = 0;
			g.seg_dst[base + k] = 0;
		}
		if (lane == 0) {
			a.out_len[i] = 0;
			if (a.status) a.status[i] = 1;
			if (a.crc) a.crc[i] = 0;
		}
		return;
	}
	uint8_t *dst = a.out + (uint64_t)i * a.out_stride;
	const int si = g.seg == HD_LAT_SEG_BYTES(1) ? 0 : g.seg == HD_LAT_SEG_BYTES(2) ? 1 : g.seg == HD_SEG_BYTES ? 2 : -1;
	const uint32_t last_len = len - (nseg - 1) * g.seg;          // 1..seg
	uint32_t pos = hdr, acc = 0;
	for (uint32_t k0 = 0; k0 < g.S; k0 += 64) {
		const uint32_t k = k0 + lane;
		const bool in = k < nseg;
		const uint32_t ol = in ? g.seg_olen[base + k] : 0u;
		const uint32_t incl = wave_incl_scan(ol);
		if (k < g.S) {
			g.seg_dst[base + k] = in ? (uint64_t)i * a.out_stride + pos + (incl - ol) : 0;
			if (!in)
				g.seg_olen[base + k] = 0;
		}
		pos += readlane(incl, 63);
		// this segment's CRC with the m full segments between it and the last one appended
		uint32_t c = in ? g.seg_crc[base + k] : 0u;
		if (in && k + 1 < nseg) {
			uint32_t m = nseg - 2 - k;
			if (si >= 0)
				for (; m; m -= (m < 16 ? m : 16))
					c = crc_shift(ct->SM[si][(m < 16 ? m : 16) - 1], c);
			else
				c = crc_append_bytes(ct, c, (uint64_t)m * g.seg);
		}
		acc ^= (in && k + 1 < nseg) ? c : 0u;        // (the last segment joins below, unshifted)
	}
	acc = wave_xor_reduce(acc);
	if (lane == 0) {
		// ... then the last segment's bytes behind all of them, and its own CRC
		uint32_t crc = nseg > 1 ? (last_len == g.seg && si >= 0 ? crc_shift(ct->SM[si][0], acc) : crc_append_bytes(ct, acc, last_len))
					: 0u;
		crc ^= g.seg_crc[base + nseg - 1];
		if (!flush) {
			dst[pos] = 0x03;                                // the empty final block
			dst[pos + 1] = 0x00;
			pos += 2;
		}
		const uint32_t paylen = pos - hdr, total = pos + trl;
		const uint32_t sizefield = a.frame == HD_FRAME_BGZF ? total - 1 : paylen;
		for (uint32_t o = 0; o < hdr; o++)
			dst[o] = (uint8_t)frame_hdr_byte(a.frame, o, sizefield);
		for (uint32_t k = 0; k < trl / 2; k++) {
			const uint32_t f = frame_trl_field(a.frame, k, crc, len);
			dst[pos + 2 * k] = (uint8_t)f;
			dst[pos + 2 * k + 1] = (uint8_t)(f >> 8);
		}
		a.out_len[i] = total;
		if (a.status) a.status[i] = 0;
		if (a.crc) a.crc[i] = crc;
	}
}

__global__ __launch_bounds__(64) void k_seg_stitch(SegArgs g)
{
	if (blockIdx.x < g.count)
		seg_stitch_member(g, blockIdx.x, threadIdx.x);
}

// The stitch and the gather in ONE launch, for members of up to 64 segments (every latency batch, every 1 MiB member): one
// WAVEFRONT PER SEGMENT.  Each works out where its block's segments go by itself -- the prefix sum over <= 64 sizes is
// twenty instructions, cheaper than waiting for another kernel to have written it -- and copies its own payload; the
// block's first wavefront also folds the CRC and writes the frame, exactly as k_seg_stitch does.  No barrier, no second
// launch behind the first: 6.5 us + 8 us + the gap between them become ~9 us of a latency batch.  (A workgroup of sixteen
// wavefronts per member, wavefront 0 stitching and a barrier ahead of the gather, was measured slower than the two launches.)
__device__ __forceinline__ void seg_finish_one(const SegArgs &g, const uint8_t *slots, uint32_t stride)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t t = blockIdx.x / g.S, k = blockIdx.x % g.S;
	if (t >= g.count)
		return;
	const DeflateArgs &a = g.a;
	const CrcTables *ct = a.ct;
	const uint32_t i = g.first + t, len = a.in_len[i];
	const uint64_t base = (uint64_t)t * g.S;
	if (len <= g.limit)
		return;                                            // not a segmented block: the ordinary coding wrote its member
	const uint32_t hdr = frame_hdr_bytes(a.frame), trl = frame_trl_bytes(a.frame);
	const bool flush = a.frame == HD_FRAME_RAW_FLUSH;
	uint64_t cap = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
	if (a.frame == HD_FRAME_BGZF && cap > 65536)
		cap = 65536;
	const uint32_t nseg = HD_SEGN_COUNT(len, g.seg);
	const bool in = lane < nseg;                           // lane = segment (g.S <= 64: the launch's promise)
	const bool ok = nseg <= g.S && (uint64_t)hdr + HD_SEGN_WORST((uint64_t)len, g.seg, flush) + trl <= cap &&
			!__ballot(in && g.seg_st[base + lane] != 0);
	if (!ok) {
		if (k == 0 && lane == 0) {
			a.out_len[i] = 0;
			if (a.status) a.status[i] = 1;
			if (a.crc) a.crc[i] = 0;
		}
		return;
	}
	if (k >= nseg)
		return;
	uint8_t *dst = a.out + (uint64_t)i * a.out_stride;
	const uint32_t ol = in ? g.seg_olen[base + lane] : 0u;
	const uint32_t incl = wave_incl_scan(ol);
	compact_one(slots + (base + k) * stride, readlane(ol, k), dst + hdr + readlane(incl, k) - readlane(ol, k), lane);
	if (k)
		return;
	// the block's first wavefront: CRC-32 of the whole from the CRCs of the parts, the frame around the payloads
	const int si = g.seg == HD_LAT_SEG_BYTES(1) ? 0 : g.seg == HD_LAT_SEG_BYTES(2) ? 1 : g.seg == HD_SEG_BYTES ? 2 : -1;
	const uint32_t last_len = len - (nseg - 1) * g.seg;          // 1..seg
	uint32_t pos = hdr + readlane(incl, 63);
	uint32_t c = in ? g.seg_crc[base + lane] : 0u;
	if (in && lane + 1 < nseg) {
		uint32_t m = nseg - 2 - lane;
		if (si >= 0)
			for (; m; m -= (m < 16 ? m : 16))
				c = crc_shift(ct->SM[si][(m < 16 ? m : 16) - 1], c);
		else
			c = crc_append_bytes(ct, c, (uint64_t)m * g.seg);
	}
	const uint32_t acc = wave_xor_reduce((in && lane + 1 < nseg) ? c : 0u);      // (the last segment joins below, unshifted)
	if (lane == 0) {
		uint32_t crc = nseg > 1 ? (last_len == g.seg && si >= 0 ? crc_shift(ct->SM[si][0], acc) : crc_append_bytes(ct, acc, last_len))
					: 0u;
		crc ^= g.seg_crc[base + nseg - 1];
		if (!flush) {
			dst[pos] = 0x03;                                // the empty final block
			dst[pos + 1] = 0x00;
			pos += 2;
		}
		const uint32_t paylen = pos - hdr, total = pos + trl;
		const uint32_t sizefield = a.frame == HD_FRAME_BGZF ? total - 1 : paylen;
		for (uint32_t o = 0; o < hdr; o++)
			dst[o] = (uint8_t)frame_hdr_byte(a.frame, o, sizefield);
		for (uint32_t q = 0; q < trl / 2; q++) {
			const uint32_t f = frame_trl_field(a.frame, q, crc, len);
			dst[pos + 2 * q] = (uint8_t)f;
			dst[pos + 2 * q + 1] = (uint8_t)(f >> 8);
		}
		a.out_len[i] = total;
		if (a.status) a.status[i] = 0;
		if (a.crc) a.crc[i] = crc;
	}
}

__global__ __launch_bounds__(64) void k_seg_finish(SegArgs g, const uint8_t *slots, uint32_t stride)
{
	seg_finish_one(g, slots, stride);
	if (g.a.done_flag) {
		// every wavefront's stores are out at system scope before it counts itself off; the last one to do so tells the host
		__threadfence_system();
		if (threadIdx.x == 0 && atomicAdd(g.a.done_count, 1u) == gridDim.x - 1) {
			__hip_atomic_store(g.a.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__threadfence_system();
			__hip_atomic_store(g.a.done_flag, g.a.done_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

// a.scratch: segmented_scratch_bytes(a.nblocks, capacity, level).  `code(args)` launches the level's
// ordinary coding of a batch (the level-1 kernel, or launch_deflate_dynamic): once for the caller's blocks
// with seg_limit set -- it takes the blocks up to the limit, the one way such a block is coded whatever
// its neighbours are -- and once per round for the segments.
template <class F>
inline int launch_deflate_segmented(const DeflateArgs &a, int level, hipStream_t st, F code)
{
	const uint64_t cap64 = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
	const uint32_t cap = cap64 > 0x7fffffffu ? 0x7fffffffu : (uint32_t)cap64;
	const uint32_t seg = a.seg_bytes, stride = seg_stride(seg);
	const uint32_t S = seg_slots_per_block(cap, seg), rb = seg_round_blocks(a.nblocks, S);
	// a.seg_limit is set: this launch takes the blocks up to the limit and leaves the longer ones alone
	int r = (a.hint & HD_HINT_NO_WHOLE) ? 0 : code(a);
	if (r || (a.hint & HD_HINT_NO_SEG))
		return r;

	uint8_t *p = a.scratch + (level >= 2 ? dynamic_scratch_bytes(a.nblocks, cap, level) : 0);
	p = (uint8_t *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
	SegArgs g;
	g.a = a;
	g.S = S;
	g.seg = seg;
	g.limit = a.seg_limit;
	const uint64_t nseg_round = (uint64_t)rb * S;
	g.seg_off = (uint64_t *)p;
	g.seg_dst = g.seg_off + nseg_round;
	g.seg_len = (uint32_t *)(g.seg_dst + nseg_round);
	g.seg_olen = g.seg_len + nseg_round;
	g.seg_crc = g.seg_olen + nseg_round;
	g.seg_st = (int32_t *)(g.seg_crc + nseg_round);
	uint8_t *slots = (uint8_t *)(((uintptr_t)(g.seg_st + nseg_round) + 15) & ~(uintptr_t)15);
	uint8_t *inner = slots + nseg_round * stride;
	inner = (uint8_t *)(((uintptr_t)inner + 15) & ~(uintptr_t)15);
	for (uint32_t first = 0; first < a.nblocks; first += rb) {
		g.first = first;
		g.count = a.nblocks - first < rb ? a.nblocks - first : rb;
		const uint32_t ns = g.count * S;
		if (a.host_seg_off && rb >= a.nblocks) {
			g.seg_off = (uint64_t *)a.host_seg_off;      // made by the host (hipdeflate_lat_run)
			g.seg_len = (uint32_t *)a.host_seg_len;
		} else {
			hipLaunchKernelGGL(k_seg_table, dim3((ns + 255) / 256), dim3(256), 0, st, g);
		}
		DeflateArgs s = a;
		s.in_off = g.seg_off;
		s.in_len = g.seg_len;
		s.nblocks = ns;
		s.frame = HD_FRAME_RAW_FLUSH;
		s.out = slots;
		s.out_stride = stride;
		s.out_cap = stride;
		s.out_len = g.seg_olen;
		s.crc = g.seg_crc;
		s.status = g.seg_st;
		s.scratch = inner;
		s.split_max = stride;
		s.seg_limit = 0;
		s.parts = HD_LAT_PARTS(level, seg);             // latency segments of the dynamic levels: parsed in parts
		s.seg_slots = seg != HD_SEG_BYTES ? S : 0;      // latency segments: primed with the end of their predecessor
		if ((r = code(s)))
			return r;
		if (S <= 64) {
			SegArgs gr = g;                                        // (a copy per round: only the last round's launch tells the host)
			if (first + rb < a.nblocks)
				gr.a.done_flag = nullptr;
			hipLaunchKernelGGL(k_seg_finish, dim3(ns), dim3(64), 0, st, gr, (const uint8_t *)slots, stride);
		} else {
			hipLaunchKernelGGL(k_seg_stitch, dim3(g.count), dim3(64), 0, st, g);
			hipLaunchKernelGGL(k_compact, dim3(ns), dim3(64), 0, st, (const uint8_t *)slots, (uint64_t)stride,
					   (const uint32_t *)g.seg_olen, (const uint64_t *)g.seg_dst, ns, a.out);
		}
	}
	return 0;
}

} // namespace hd
