// hd_api.hip -- the C ABI of libhipdeflate.so (include/hipdeflate.h): context,
// device/pinned pools, kernel launches, and the per-block zlibutil-style codecs.
//
// No CPU codec lives here: when no gfx950 device is usable every entry point
// fails loudly (HD_E_NODEVICE + one line on stderr).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/hipdeflate.h"
#include "hd_deflate_static.hpp"
#include "hd_deflate_dynamic.hpp"
#include "hd_deflate_wg.hpp"
#include "hd_emit_wg.hpp"
#include "hd_inflate.hpp"
#include "hd_inflate_lat.hpp"
#include "hd_compact.hpp"
#include "hd_segment.hpp"

namespace {

using hd::CrcTables;

#define HD_CHECK(expr)                                                                         \
	do {                                                                                   \
		hipError_t e_ = (expr);                                                        \
		if (e_ != hipSuccess) {                                                        \
			fprintf(stderr, "hipdeflate: %s failed: %s (%s:%d)\n", #expr,          \
				hipGetErrorString(e_), __FILE__, __LINE__);                    \
			return e_ == hipErrorOutOfMemory ? HD_E_NOMEM : HD_E_NODEVICE;         \
		}                                                                              \
	} while (0)

struct Buf {
	void *p = nullptr;
	size_t cap = 0;
	bool pinned = false;
	int reserve(size_t n)
	{
		if (n <= cap)
			return 0;
		free_current();
		size_t want = n + n / 4 + 4096;
		hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
		if (e != hipSuccess) {
			p = nullptr;
			cap = 0;
			fprintf(stderr, "hipdeflate: %s of %zu bytes failed: %s\n", pinned ? "hipHostMalloc" : "hipMalloc",
				want, hipGetErrorString(e));
			return HD_E_NOMEM;
		}
		cap = want;
		return 0;
	}
	void free_current()
	{
		if (p)
			(void)(pinned ? hipHostFree(p) : hipFree(p));
		p = nullptr;
		cap = 0;
	}
	void release()
	{
		free_current();
		for (void *q : retired)
			(void)(pinned ? hipHostFree(q) : hipFree(q));
		retired.clear();
	}
	// grow while launches on other streams may still use the old allocation: it is kept until release() instead of
	// being freed behind a device-wide synchronisation (a grow-only buffer retires a handful of allocations at most)
	std::vector<void *> retired;
	int grow_keep_old(size_t n)
	{
		if (n <= cap)
			return 0;
		void *old = p;
		const size_t old_cap = cap;
		p = nullptr;
		cap = 0;
		const int r = reserve(n);
		if (r) {                                   // (nothing changed: the caller may settle for less -- batch_deflate_dev_impl)
			p = old;
			cap = old_cap;
			return r;
		}
		if (old)
			retired.push_back(old);
		return r;
	}
};

struct Ctx {
	bool ready = false;
	int failed = 0;
	int device = -1;
	char desc[256] = "hipdeflate (not initialised)";
	hipStream_t stream = nullptr;
	CrcTables *d_ct = nullptr;
	uint32_t *d_stalls = nullptr;    // blocks the workgroup parse gave up on (hipdeflate_stall_count)
	// host-pointer API pools (guarded by mu)
	std::mutex mu;
	Buf d_in, d_meta, d_slots, d_packed, d_scratch, d_scan;
	Buf h_in{ nullptr, 0, true }, h_meta{ nullptr, 0, true }, h_out{ nullptr, 0, true };
	// device-pointer API scratch (token slabs for the dynamic levels), guarded by mu_dev
	std::mutex mu_dev;
	Buf d_tok;
	Buf d_tiles;
	hipEvent_t ev_tok = nullptr;     // last launch that used d_tok: a launch on ANOTHER stream waits for it
	hipStream_t st_tok = nullptr;
	bool tok_used = false;
	hd::WgBeside beside;             // the workgroup levels' emit kernel beside their parse (hd_deflate_wg.hpp), with d_tok
	hipEvent_t ev_tiles = nullptr;   // the same for d_tiles
	hipStream_t st_tiles = nullptr;
	bool tiles_used = false;
};

// One context per entry of the device list (hipdeflate_init(devices...)).  An entry is a HIP device ordinal;
// the same ordinal may appear twice (HIPDEFLATE_DEVICES=0,0: two independent contexts on one card -- how the multi-device
// hosts are rehearsed on a one-GPU box).  A thread works on the entry hipdeflate_use_device() chose for it (default: entry
// 0); pipes and latency contexts keep the entry they were opened on.
constexpr int HD_MAX_CTX = 32;
Ctx g_all[HD_MAX_CTX];
int g_want[HD_MAX_CTX];              // the device ordinal asked for per entry (-1: from the environment)
int g_nctx = 0;                      // entries configured (0: not yet -- the first use reads the environment)
thread_local int t_cur = 0;
std::mutex g_init_mu;
inline Ctx &cur() { return g_all[t_cur]; }

} // namespace
int hd_probe_lds_order(void);        // hd_selftest.hip
namespace {
// the per-block codecs' pool of latency contexts (deflate_one below), closed by hipdeflate_shutdown()
constexpr int CODEC_LEVELS = 13;                          // levels 0..12 (the kernels clamp above 9)
constexpr size_t CODEC_POOL_MAX = 64;                     // idle contexts kept per level (~200 KiB pinned each)
std::mutex g_codec_mu;
std::vector<hipdeflate_lat *> g_codec_free[CODEC_LEVELS];
void codec_pool_drain()
{
	std::lock_guard<std::mutex> lk(g_codec_mu);
	for (std::vector<hipdeflate_lat *> &f : g_codec_free) {
		for (hipdeflate_lat *c : f)
			hipdeflate_lat_close(c);
		f.clear();
	}
}

void build_crc_tables(CrcTables *t)
{
	const uint32_t poly = 0xEDB88320u;
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for (int k = 0; k < 8; k++)
			c = (c >> 1) ^ (poly & (0u - (c & 1u)));
		t->T[0][i] = c;
	}
	for (int k = 1; k < 4; k++)
		for (uint32_t i = 0; i < 256; i++)
			t->T[k][i] = (t->T[k - 1][i] >> 8) ^ t->T[0][t->T[k - 1][i] & 0xff];
	// B[k][v]: state (v << 8k) after 1024 zero bytes; B16: after 16; BL: after 1008
	for (int k = 0; k < 4; k++)
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t s = i << (8 * k);
			for (int z = 0; z < 1024; z++) {
				if (z == 16)
					t->B16[k][i] = s;
				if (z == 1008)
					t->BL[k][i] = s;
				s = t->T[0][s & 0xff] ^ (s >> 8);
			}
			t->B[k][i] = s;
		}
	// T16[j][b]: byte b followed by j zero bytes (slicing-by-16)
	for (uint32_t i = 0; i < 256; i++)
		t->T16[0][i] = t->T[0][i];
	for (int j = 1; j < 16; j++)
		for (uint32_t i = 0; i < 256; i++)
			t->T16[j][i] = (t->T16[j - 1][i] >> 8) ^ t->T[0][t->T16[j - 1][i] & 0xff];
	// SL: static litlen codes (RFC 1951 3.2.6), bit-reversed, length extra bits appended, bit count << 16
	{
		auto rev = [](uint32_t c, int bits) {
			uint32_t r = 0;
			for (int i = 0; i < bits; i++)
				r |= ((c >> i) & 1u) << (bits - 1 - i);
			return r;
		};
		auto litlen = [&](uint32_t sym, uint32_t &code, uint32_t &bits) {
			if (sym < 144) { code = rev(0x30 + sym, 8); bits = 8; }
			else if (sym < 256) { code = rev(0x190 + (sym - 144), 9); bits = 9; }
			else if (sym < 280) { code = rev(sym - 256, 7); bits = 7; }
			else { code = rev(0xC0 + (sym - 280), 8); bits = 8; }
		};
		static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
						    67, 83, 99, 115, 131, 163, 195, 227, 258 };
		static const uint8_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
						  4, 4, 4, 4, 5, 5, 5, 5, 0 };
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t c, b;
			litlen(i, c, b);
			t->SL[i] = c | (b << 16);
		}
		for (uint32_t len = 3; len <= 258; len++) {
			uint32_t slot = 28;
			while (lbase[slot] > len)
				slot--;
			uint32_t c, b;
			litlen(257 + slot, c, b);
			t->SL[256 + len - 3] = (c | ((len - lbase[slot]) << b)) | ((b + lext[slot]) << 16);
		}
	}
	// P2[j]: 2^j zero bytes appended (each the square of the one before); SM[s][m-1]: m segments' worth
	{
		auto apply = [](const uint32_t (*Z)[256], uint32_t v) {
			return Z[0][v & 0xff] ^ Z[1][(v >> 8) & 0xff] ^ Z[2][(v >> 16) & 0xff] ^ Z[3][v >> 24];
		};
		for (int k = 0; k < 4; k++)
			for (uint32_t i = 0; i < 256; i++) {
				const uint32_t v = i << (8 * k);
				t->P2[0][k][i] = t->T[0][v & 0xff] ^ (v >> 8);
			}
		for (int j = 1; j < 24; j++)
			for (int k = 0; k < 4; k++)
				for (uint32_t i = 0; i < 256; i++)
					t->P2[j][k][i] = apply(t->P2[j - 1], t->P2[j - 1][k][i]);
		static const uint32_t segs[3] = { HD_LAT_SEG_BYTES(1), HD_LAT_SEG_BYTES(2), HD_SEG_BYTES };
		for (int si = 0; si < 3; si++) {
			for (int k = 0; k < 4; k++)
				for (uint32_t i = 0; i < 256; i++) {
					uint32_t v = i << (8 * k);
					for (int j = 0; j < 24; j++)
						if (segs[si] & (1u << j))
							v = apply(t->P2[j], v);
					t->SM[si][0][k][i] = v;
				}
			for (int m = 1; m < 16; m++)                 // (the one-segment operator is complete by now)
				for (int k = 0; k < 4; k++)
					for (uint32_t i = 0; i < 256; i++)
						t->SM[si][m][k][i] = apply(t->SM[si][0], t->SM[si][m - 1][k][i]);
		}
	}
	// K[q] = x^(128 q): appending 16 q zero bytes to the state 0x80000000 (= x^0)
	uint32_t s = 0x80000000u;
	for (int q = 0; q < 64; q++) {
		t->K[q] = s;
		for (int z = 0; z < 16; z++)
			s = t->T[0][s & 0xff] ^ (s >> 8);
	}
}

// (g_init_mu held) the device list, once: an explicit list, or the environment -- HIPDEFLATE_DEVICES=0,1,... (a list),
// HIPDEFLATE_DEVICE / LOCAL_RANK (one ordinal: a torch.distributed rank owns one card), else device 0
void configure_locked(const int *devices, int n)
{
	if (g_nctx)
		return;
	if (devices && n > 0) {
		for (int i = 0; i < n && i < HD_MAX_CTX; i++)
			g_want[g_nctx++] = devices[i];
		return;
	}
	const char *s = getenv("HIPDEFLATE_DEVICES");
	if (s && *s) {
		while (*s && g_nctx < HD_MAX_CTX) {
			char *end = nullptr;
			const long v = strtol(s, &end, 10);
			if (end == s)
				break;
			g_want[g_nctx++] = (int)(v < 0 ? 0 : v);
			s = *end == ',' ? end + 1 : end;
		}
	}
	if (!g_nctx)
		g_want[g_nctx++] = -1;
}

int ctx_init(Ctx &g, int device)
{
	std::lock_guard<std::mutex> lk(g_init_mu);
	if (g.ready)
		return 0;
	if (g.failed)
		return g.failed;
	// HIPDEFLATE_INIT_TRACE=1: where the start-up goes, one line on stderr (profiles/r05_startup.txt)
	const bool trace = getenv("HIPDEFLATE_INIT_TRACE") != nullptr;
	double ts[8];
	int nts = 0;
	auto stamp = [&]() {
		if (trace && nts < 8) {
			struct timespec t;
			clock_gettime(CLOCK_MONOTONIC, &t);
			ts[nts++] = t.tv_sec * 1e3 + t.tv_nsec / 1e6;
		}
	};
	stamp();
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	stamp();
	if (e != hipSuccess || ndev == 0) {
		fprintf(stderr, "hipdeflate: no HIP device (%s); there is no CPU fallback\n",
			e == hipSuccess ? "device count 0" : hipGetErrorString(e));
		return g.failed = HD_E_NODEVICE;
	}
	if (device < 0) {
		const char *s = getenv("HIPDEFLATE_DEVICE");
		if (!s || !*s)
			s = getenv("LOCAL_RANK");
		device = s && *s ? atoi(s) % ndev : 0;
	}
	if (device >= ndev)
		device = device % ndev;
	hipDeviceProp_t prop;
	HD_CHECK(hipSetDevice(device));
	HD_CHECK(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		fprintf(stderr, "hipdeflate: device %d is %s; this library holds gfx950 (MI355X) code only\n", device,
			prop.gcnArchName);
		return g.failed = HD_E_NODEVICE;
	}
	stamp();
	HD_CHECK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
	stamp();
	CrcTables *h = (CrcTables *)malloc(sizeof(CrcTables));
	build_crc_tables(h);
	stamp();
	HD_CHECK(hipMalloc((void **)&g.d_ct, sizeof(CrcTables)));
	HD_CHECK(hipMemcpy(g.d_ct, h, sizeof(CrcTables), hipMemcpyHostToDevice));
	free(h);
	HD_CHECK(hipMalloc((void **)&g.d_stalls, 16));
	HD_CHECK(hipMemset(g.d_stalls, 0, 16));
	stamp();
	// Several lanes' ds_write_b16 to one table entry in one instruction: the highest lane's data must stay (the parse
	// kernels and their CPU twin lean on it, hd_deflate_static.hpp fetch()).  Probed once per context: on a device
	// that arbitrates differently the streams would still be valid DEFLATE but not the twin's bytes, and ranks of one
	// job could disagree -- refused rather than run.
	if (const int bad = hd_probe_lds_order()) {
		fprintf(stderr, "hipdeflate: device %d failed the LDS store-order probe (%d); this build's encoders are not "
			"reproducible on it and refuse to run\n", device, bad);
		(void)hipFree(g.d_ct);
		g.d_ct = nullptr;
		(void)hipFree(g.d_stalls);
		g.d_stalls = nullptr;
		(void)hipStreamDestroy(g.stream);
		g.stream = nullptr;
		return g.failed = HD_E_NODEVICE;
	}
	stamp();
	if (trace && nts == 7)
		fprintf(stderr, "hipdeflate init (ms): runtime + device count %.1f, set device + properties %.1f, stream %.1f, CRC tables on the CPU %.1f, "
			"first allocations + copy %.1f, first kernel (code object load + LDS order probe) %.1f; total %.1f\n",
			ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[6] - ts[0]);
	g.device = device;
	snprintf(g.desc, sizeof(g.desc), "hipdeflate 0.2 on device %d: %s (%s), %d CUs, %.0f GiB", device, prop.name,
		 prop.gcnArchName, prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0);
	g.ready = true;
	return 0;
}

// entry `idx` of the device list, created on first use
inline int ensure_ctx(int idx)
{
	Ctx &g = g_all[idx];
	if (g.ready)
		return 0;
	{
		std::lock_guard<std::mutex> lk(g_init_mu);
		configure_locked(nullptr, 0);
		if (idx >= g_nctx)
			return HD_E_ARG;
	}
	return ctx_init(g, g_want[idx]);
}
// the calling thread's context
inline int ensure() { return ensure_ctx(t_cur); }

// the host-pointer entry points may be called from any thread; the device of
// the calling thread must be the context's
inline int bind_device(const Ctx &g)
{
	HD_CHECK(hipSetDevice(g.device));
	return 0;
}
inline int bind_device() { return bind_device(cur()); }

// the methods of a pipe / latency context run on the entry it was opened on, whatever entry the calling thread has chosen
struct OnCtx {
	int keep;
	explicit OnCtx(int idx) : keep(t_cur) { t_cur = idx; }
	~OnCtx() { t_cur = keep; }
};

// the ordinary coding of a batch at `level`: one wave per block
static int code_batch(const hd::DeflateArgs &a, int level, hipStream_t st)
{
	if (level <= 1) {
		hd::DeflateArgs b = a;
		b.level = level;
		if (a.seg_slots)                    // latency segments: the instantiation with priming compiled in
			hipLaunchKernelGGL((hd::k_deflate_static<HD_L1_WIN_BITS, HD_L1_HASH_BITS, false, HD_MIN_MATCH, 0, 0, 0, true>), dim3(a.nblocks),
					   dim3(64), 0, st, b);
		else
			hipLaunchKernelGGL((hd::k_deflate_static<HD_L1_WIN_BITS, HD_L1_HASH_BITS, false>), dim3(a.nblocks), dim3(64), 0, st, b);
		return 0;
	}
	return hd::launch_deflate_dynamic(a, level, st);
}

int launch_deflate(const hd::DeflateArgs &a, int level, hipStream_t st)
{
	if (a.nblocks == 0)
		return 0;
	int r = a.seg_limit
			? hd::launch_deflate_segmented(a, level, st, [&](const hd::DeflateArgs &x) { return code_batch(x, level, st); })
			: code_batch(a, level, st);
	if (r)
		return r;
	if (a.frame == HD_FRAME_ZLIB)
		hipLaunchKernelGGL(hd::k_adler32_patch, dim3(a.nblocks), dim3(64), 0, st, a.in, a.in_off, a.in_len, a.nblocks,
				   a.out, a.out_stride, a.out_len, a.status);
	HD_CHECK(hipGetLastError());
	return 0;
}

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

} // namespace

extern "C" {

int hipdeflate_init(int device)
{
	{
		// device < 0 = "from the environment": HIPDEFLATE_DEVICES (a list) if set, else HIPDEFLATE_DEVICE / LOCAL_RANK /
		// 0 -- the same list every other entry point would configure lazily (-1 used to pin a list of one
		// and HIPDEFLATE_DEVICES was never read by the hosts that call this first)
		std::lock_guard<std::mutex> lk(g_init_mu);
		if (device < 0)
			configure_locked(nullptr, 0);
		else
			configure_locked(&device, 1);
	}
	return ensure();
}

int hipdeflate_init_devices(const int *devices, int n)
{
	if (!devices || n < 1 || n > HD_MAX_CTX)
		return HD_E_ARG;
	{
		std::lock_guard<std::mutex> lk(g_init_mu);
		if (g_nctx) {                                     // a list is already in force: the same one is fine
			if (g_nctx != n)
				return HD_E_ARG;
			for (int i = 0; i < n; i++)
				if (g_want[i] != devices[i] && !(g_all[i].ready && g_all[i].device == devices[i]))
					return HD_E_ARG;
		}
		configure_locked(devices, n);
	}
	const int keep = t_cur;
	int r = 0;
	for (int i = 0; i < n && !r; i++) {
		t_cur = i;
		r = ensure();
	}
	t_cur = keep;
	return r;
}

int hipdeflate_device_count(void)
{
	std::lock_guard<std::mutex> lk(g_init_mu);
	configure_locked(nullptr, 0);
	return g_nctx;
}

int hipdeflate_use_device(int index)
{
	if (index < 0 || index >= hipdeflate_device_count())
		return HD_E_ARG;
	t_cur = index;
	return ensure();
}

int hipdeflate_available(void) { return ensure(); }

uint64_t hipdeflate_stall_count(void)
{
	uint64_t sum = 0;
	std::lock_guard<std::mutex> lk(g_init_mu);
	for (int i = 0; i < HD_MAX_CTX; i++) {
		Ctx &g = g_all[i];
		if (!g.ready || !g.d_stalls)
			continue;
		uint32_t v = 0;
		if (hipSetDevice(g.device) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
		    hipMemcpy(&v, g.d_stalls, 4, hipMemcpyDeviceToHost) == hipSuccess)
			sum += v;
	}
	return sum;
}

const char *hipdeflate_version(void) { return cur().desc; }

static void infb_drain();

void hipdeflate_shutdown(void)
{
	codec_pool_drain();                                  // the per-block codecs' contexts (their own locks)
	infb_drain();
	std::lock_guard<std::mutex> lk(g_init_mu);
	for (int i = 0; i < HD_MAX_CTX; i++) {
		Ctx &g = g_all[i];
		g.failed = 0;
		if (!g.ready)
			continue;
		(void)hipSetDevice(g.device);
		(void)hipDeviceSynchronize();                        // launches on callers' streams may still use our scratch
		for (Buf *b : { &g.d_in, &g.d_meta, &g.d_slots, &g.d_packed, &g.d_scratch, &g.d_scan, &g.h_in, &g.h_meta,
				&g.h_out, &g.d_tok, &g.d_tiles })
			b->release();
		(void)hipFree(g.d_ct);
		(void)hipFree(g.d_stalls);
		g.d_stalls = nullptr;
		(void)hipStreamDestroy(g.stream);
		if (g.ev_tok)
			(void)hipEventDestroy(g.ev_tok);
		if (g.ev_tiles)
			(void)hipEventDestroy(g.ev_tiles);
		g.ev_tok = g.ev_tiles = nullptr;
		g.beside.release();
		g.st_tok = g.st_tiles = nullptr;
		g.tok_used = g.tiles_used = false;
		g.d_ct = nullptr;
		g.stream = nullptr;
		g.ready = false;
	}
	g_nctx = 0;                                          // the next use configures again (environment or an explicit list)
}

static uint64_t scratch_need(uint32_t nblocks, uint32_t cap, int level, bool latency)
{
	if (level >= 1 && level < HD_WG_LEVEL && latency && cap > HD_LAT_SEG_BYTES(level))      // (the workgroup levels have no segments in any mode)
		return hd::segmented_scratch_bytes(nblocks, cap, level, HD_LAT_SEG_BYTES(level));
	if (level >= 1 && level < HD_WG_LEVEL && cap > HD_SEG_LIMIT)       // (the workgroup levels take a block of any length whole)
		return hd::segmented_scratch_bytes(nblocks, cap, level);
	return level < 2 ? 0 : hd::dynamic_scratch_bytes(nblocks, cap, level, 0, latency);
}

uint64_t hipdeflate_scratch_bytes(uint32_t nblocks, uint32_t max_block, int level)
{
	return scratch_need(nblocks, max_block, level, false);
}

uint64_t hipdeflate_bound(uint64_t block_bytes, int level)
{
	// (latency mode included: its segments are the smallest, their worst case the largest)
	const uint32_t lat = HD_LAT_SEG_BYTES(level);
	const uint64_t payload = (level >= 1 && level < HD_WG_LEVEL && block_bytes > lat) ? HD_SEGN_WORST(block_bytes, lat, 0)
								   : block_bytes + 5 * (block_bytes / 65535 + 1) + 5;
	return (payload + 32 + 15) & ~(uint64_t)15;       // + the longest container (20 + 8 bytes)
}

/* ---- device-pointer API ---------------------------------------------------- */

// max_in: the longest block of the batch where the host knows the lengths (0: only the device does)
// The emit kernel BESIDE the parse (hd_deflate_wg.hpp launch_wg) needs its two kernels on the device at the same time.  Where the
// process says that kernels run one at a time -- the runtime's launch-blocking and serialising switches, a profiler collecting
// hardware counters (rocprofv3 --pmc dispatches one kernel at a time and exports these variables to its child) -- the emit kernel
// follows the parse as in round 4: the same bytes, and no resident wavefronts waiting (bounded, ~2 s a block) for a parse that
// cannot start.  HIPDEFLATE_NO_BESIDE=1 says the same by hand.
static uint32_t g_test_beside_keep = 3;     // hipdeflate_test_beside
static bool beside_allowed()
{
	auto on = [](const char *name) {
		const char *v = getenv(name);
		return v && *v && strcmp(v, "0") != 0;
	};
	// (read at every launch -- a launch is >= 512 blocks --, so a test can switch it)
	return !on("HIPDEFLATE_NO_BESIDE") && !on("HIP_LAUNCH_BLOCKING") && !on("AMD_SERIALIZE_KERNEL") && !on("ROCPROF_COUNTER_COLLECTION") &&
	       !getenv("ROCPROF_COUNTERS");
}

static int batch_deflate_dev_impl(const void *in, const void *in_off, const void *in_len, uint32_t nblocks, int level,
				  int frame, void *out, uint64_t out_stride, uint32_t out_cap, void *out_len,
				  void *crc32, void *status, void *stream, uint32_t max_in)
{
	Ctx &g = cur();
	int r = ensure();
	if (r)
		return r;
	const bool latency = (frame & HD_FRAME_LATENCY) != 0;
	frame &= ~HD_FRAME_LATENCY;
	if (frame < HD_FRAME_RAW || frame > HD_FRAME_GZIP || (out_stride & 15) || ((uintptr_t)out & 15) || !out_len)
		return HD_E_ARG;
	hd::DeflateArgs a;
	a.in = (const uint8_t *)in;
	a.in_off = (const uint64_t *)in_off;
	a.in_len = (const uint32_t *)in_len;
	a.nblocks = nblocks;
	a.frame = frame;
	a.level = level;
	a.out = (uint8_t *)out;
	a.out_stride = out_stride;
	a.out_cap = out_cap;
	a.out_len = (uint32_t *)out_len;
	a.crc = (uint32_t *)crc32;
	a.status = (int32_t *)status;
	a.ct = g.d_ct;
	a.scratch = nullptr;
	a.first = 0;
	a.count = 0;
	a.skip_small = 0;
	a.split_max = hd::split_max_block(out_stride, out_cap);
	a.split_ovf = nullptr;
	// slots that could hold a block longer than the segment limit (HD_SEG_LIMIT; the segment size itself in latency
	// mode): such blocks are coded in segments, the ordinary coding leaves them alone
	const uint32_t seg_lim = latency ? HD_LAT_SEG_BYTES(level) : HD_SEG_LIMIT;
	a.seg_bytes = latency ? HD_LAT_SEG_BYTES(level) : HD_SEG_BYTES;
	a.seg_limit = (level >= 1 && level < HD_WG_LEVEL && a.split_max > seg_lim) ? seg_lim : 0;
	// the workgroup levels: the parse's records are sized by the longest block where the host knows it, else by the slot
	// (k_parse_wg refuses a block longer than its record)
	if (level >= HD_WG_LEVEL && max_in)
		a.split_max = max_in;
	// ... and in latency mode the member is written by a workgroup, the same bytes (blocks up to 64 KiB)
	a.lat = (latency && level >= HD_WG_LEVEL && a.split_max <= hd::EW_BLOCK_MAX) ? 1u : 0u;
	a.stalls = g.d_stalls;
	a.hint = 0;
	a.host_seg_off = nullptr;
	a.host_seg_len = nullptr;
	const uint64_t need = scratch_need(nblocks, a.split_max, level, latency);
	if (need) {
		// token slabs of the dynamic levels, segment slots of large blocks: library-owned, grow-only
		std::lock_guard<std::mutex> lk(g.mu_dev);
		// (a re-allocation must not pull the rug from under launches in flight: the old scratch stays alive)
		bool beside_fits = true;
		if (g.d_tok.grow_keep_old(need)) {
			// (what the emit kernel BESIDE the parse adds -- a second buffer of records, the flag lines -- is an option, not a need:
			// without it the launch runs in the old order)
			const uint64_t extra = (level >= HD_WG_LEVEL && !a.lat) ? hd::wg_beside_bytes(nblocks, a.split_max) : 0;
			if (!extra || extra >= need || g.d_tok.grow_keep_old(need - extra))
				return HD_E_NOMEM;
			beside_fits = false;
		}
		a.scratch = (uint8_t *)g.d_tok.p;
		// the slabs are shared by every launch: launches on different streams take turns
		if (!g.ev_tok)
			HD_CHECK(hipEventCreateWithFlags(&g.ev_tok, hipEventDisableTiming));
		if (g.tok_used && g.st_tok != (hipStream_t)stream)          // same stream: already in order
			HD_CHECK(hipStreamWaitEvent((hipStream_t)stream, g.ev_tok, 0));
		if (level >= HD_WG_LEVEL && !a.lat && beside_fits && beside_allowed() && g.beside.init() == 0) {
			a.beside = &g.beside;
			// (emit wavefronts a CU keeps: three -- but two beside the four-way parse of BGZF-sized blocks, where the third costs the parse
			// more than it takes off the launch's end: encode_l6 123.8 -> 125.8 GB/s, 1 MiB members the other way, 131.1 -> 128.5;
			// tools/r05_keep_ab.sh)
			const uint32_t keep = (HD_WG_WAYS(level) == 4 && a.split_max <= 65536) ? 2u : 3u;
			a.beside_keep = g_test_beside_keep < keep ? g_test_beside_keep : keep;
		}
		r = launch_deflate(a, level, (hipStream_t)stream);
		if (!r) {
			HD_CHECK(hipEventRecord(g.ev_tok, (hipStream_t)stream));
			g.st_tok = (hipStream_t)stream;
			g.tok_used = true;
		}
		return r;
	}
	return launch_deflate(a, level, (hipStream_t)stream);
}

int hipdeflate_batch_deflate_dev(const void *in, const void *in_off, const void *in_len, uint32_t nblocks, int level,
				 int frame, void *out, uint64_t out_stride, uint32_t out_cap, void *out_len,
				 void *crc32, void *status, void *stream)
{
	return batch_deflate_dev_impl(in, in_off, in_len, nblocks, level, frame, out, out_stride, out_cap, out_len, crc32, status,
				      stream, 0);
}

static int batch_inflate_dev(const void *in, const void *in_off, const void *in_len, uint32_t nblocks, void *out,
			     const void *out_off, const void *out_cap, void *out_len, void *crc32, void *status, void *stream,
			     uint32_t flags)
{
	Ctx &g = cur();
	int r = ensure();
	if (r)
		return r;
	if (!out_len)
		return HD_E_ARG;
	if (nblocks == 0)
		return 0;
	hd::InflateArgs a;
	a.in = (const uint8_t *)in;
	a.in_off = (const uint64_t *)in_off;
	a.in_len = (const uint32_t *)in_len;
	a.nblocks = nblocks;
	a.out = (uint8_t *)out;
	a.out_off = (const uint64_t *)out_off;
	a.out_cap = (const uint32_t *)out_cap;
	a.out_len = (uint32_t *)out_len;
	a.crc = (uint32_t *)crc32;
	a.status = (int32_t *)status;
	a.ct = g.d_ct;
	a.flags = flags;
	hipLaunchKernelGGL(hd::k_inflate, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, a);
	HD_CHECK(hipGetLastError());
	return 0;
}

int hipdeflate_batch_inflate_dev(const void *in, const void *in_off, const void *in_len, uint32_t nblocks, void *out,
				 const void *out_off, const void *out_cap, void *out_len, void *crc32, void *status,
				 void *stream)
{
	return batch_inflate_dev(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, crc32, status, stream, 0);
}

int hipdeflate_batch_inflate_flush_dev(const void *in, const void *in_off, const void *in_len, uint32_t nblocks, void *out,
				       const void *out_off, const void *out_cap, void *out_len, void *crc32, void *status,
				       void *stream)
{
	return batch_inflate_dev(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, crc32, status, stream,
				 hd::INF_FLUSHED);
}

int hipdeflate_scan_sizes_dev(const void *out_len, uint32_t nblocks, uint64_t base, void *dst_off, void *total,
			      void *stream)
{
	Ctx &g = cur();
	int r = ensure();
	if (r)
		return r;
	if (nblocks == 0) {
		if (total)
			HD_CHECK(hipMemsetAsync(total, 0, 8, (hipStream_t)stream));
		return 0;
	}
	const uint32_t ntiles = (nblocks + hd::SCAN_TILE - 1) / hd::SCAN_TILE;
	// the tile buffer is one for the whole library: scans on different streams take turns
	std::lock_guard<std::mutex> lk(g.mu_dev);
	if (g.d_tiles.grow_keep_old((size_t)ntiles * 8))
		return HD_E_NOMEM;
	uint64_t *tiles = (uint64_t *)g.d_tiles.p;
	hipStream_t st = (hipStream_t)stream;
	if (!g.ev_tiles)
		HD_CHECK(hipEventCreateWithFlags(&g.ev_tiles, hipEventDisableTiming));
	if (g.tiles_used && g.st_tiles != st)                               // same stream: already in order
		HD_CHECK(hipStreamWaitEvent(st, g.ev_tiles, 0));
	hipLaunchKernelGGL(hd::k_scan_tile_sums, dim3(ntiles), dim3(256), 0, st, (const uint32_t *)out_len, nblocks, tiles);
	hipLaunchKernelGGL(hd::k_scan_tiles, dim3(1), dim3(256), 0, st, tiles, ntiles, base, (uint64_t *)total);
	hipLaunchKernelGGL(hd::k_scan_finish, dim3(ntiles), dim3(256), 0, st, (const uint32_t *)out_len, nblocks, tiles,
			   (uint64_t *)dst_off);
	HD_CHECK(hipGetLastError());
	HD_CHECK(hipEventRecord(g.ev_tiles, st));
	g.st_tiles = st;
	g.tiles_used = true;
	return 0;
}

int hipdeflate_compact_dev(const void *slots, uint64_t stride, const void *out_len, const void *dst_off,
			   uint32_t nblocks, void *dst, void *stream)
{
	int r = ensure();
	if (r)
		return r;
	if (nblocks == 0)
		return 0;
	if ((stride & 3) || ((uintptr_t)slots & 3))
		return HD_E_ARG;
	hipLaunchKernelGGL(hd::k_compact, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)slots, stride,
			   (const uint32_t *)out_len, (const uint64_t *)dst_off, nblocks, (uint8_t *)dst);
	HD_CHECK(hipGetLastError());
	return 0;
}

int hipdeflate_compact_span_dev(const void *slots, uint64_t stride, const void *out_len, const void *dst_off,
				uint32_t nblocks, void *span, uint64_t span_base, void *stream)
{
	// k_compact only ever forms dst + dst_off[i] with dst_off[i] >= span_base: the address of "stream byte 0"
	return hipdeflate_compact_dev(slots, stride, out_len, dst_off, nblocks, (uint8_t *)span - span_base, stream);
}

/* ---- host-pointer API ------------------------------------------------------ */

int hipdeflate_batch_deflate(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks,
			     int level, int frame, uint8_t *out, uint64_t out_stride, uint32_t out_cap,
			     uint32_t *out_len, uint32_t *crc32, int32_t *status)
{
	Ctx &g = cur();
	int r = ensure();
	if (r)
		return r;
	if (nblocks == 0)
		return 0;
	if (!in_off || !in_len || !out || !out_len || (frame & ~HD_FRAME_LATENCY) < HD_FRAME_RAW ||
	    (frame & ~HD_FRAME_LATENCY) > HD_FRAME_GZIP)
		return HD_E_ARG;
	std::lock_guard<std::mutex> lk(g.mu);
	if ((r = bind_device(g)))
		return r;

	// pack the blocks 16-byte aligned into pinned memory (one H2D, aligned fast path)
	size_t in_total = 0, max_len = 0;
	for (uint32_t i = 0; i < nblocks; i++) {
		in_total += up16(in_len[i]);
		if (in_len[i] > max_len)
			max_len = in_len[i];
	}
	const uint64_t cap_user = out_stride < out_cap ? out_stride : out_cap;
	// device slot: what the user allows, but never more than any encoding needs
	const size_t need = (size_t)hipdeflate_bound(max_len, level);
	const size_t slot = up16(cap_user < need ? cap_user : need);
	const size_t meta_bytes = (size_t)nblocks * (8 + 4 + 4 + 4 + 4 + 8);
	if (g.h_in.reserve(in_total + 16) || g.h_meta.reserve(meta_bytes) || g.d_in.reserve(in_total + 16) ||
	    g.d_meta.reserve(meta_bytes) || g.d_slots.reserve(slot * nblocks + 16) || g.d_scan.reserve(16))
		return HD_E_NOMEM;
	uint8_t *hin = (uint8_t *)g.h_in.p;
	uint64_t *h_off = (uint64_t *)g.h_meta.p;
	uint32_t *h_len = (uint32_t *)(h_off + nblocks);
	size_t o = 0;
	for (uint32_t i = 0; i < nblocks; i++) {
		h_off[i] = o;
		h_len[i] = in_len[i];
		if (in_len[i])
			memcpy(hin + o, in + in_off[i], in_len[i]);
		o += up16(in_len[i]);
	}
	uint8_t *dm = (uint8_t *)g.d_meta.p;
	uint64_t *d_off = (uint64_t *)dm;
	uint32_t *d_len = (uint32_t *)(d_off + nblocks);
	uint32_t *d_olen = d_len + nblocks;
	uint32_t *d_crc = d_olen + nblocks;
	int32_t *d_st = (int32_t *)(d_crc + nblocks);
	uint64_t *d_doff = (uint64_t *)(d_st + nblocks);
	HD_CHECK(hipMemcpyAsync(g.d_in.p, hin, in_total, hipMemcpyHostToDevice, g.stream));
	HD_CHECK(hipMemcpyAsync(d_off, h_off, (size_t)nblocks * 12, hipMemcpyHostToDevice, g.stream));
	r = batch_deflate_dev_impl(g.d_in.p, d_off, d_len, nblocks, level, frame, g.d_slots.p, slot,
				   (uint32_t)(cap_user < slot ? cap_user : slot), d_olen, d_crc, d_st, g.stream,
				   max_len ? (uint32_t)max_len : 1u);
	if (r)
		return r;
	// gather on the device so that only the compressed bytes cross PCIe
	uint64_t *d_total = (uint64_t *)g.d_scan.p;
	if ((r = hipdeflate_scan_sizes_dev(d_olen, nblocks, 0, d_doff, d_total, g.stream)))
		return r;
	if (g.d_packed.reserve(slot * nblocks + 16))
		return HD_E_NOMEM;
	if ((r = hipdeflate_compact_dev(g.d_slots.p, slot, d_olen, d_doff, nblocks, g.d_packed.p, g.stream)))
		return r;
	uint32_t *h_olen = (uint32_t *)g.h_meta.p;               // reuse: olen, crc, status
	HD_CHECK(hipMemcpyAsync(h_olen, d_olen, (size_t)nblocks * 12, hipMemcpyDeviceToHost, g.stream));
	uint64_t total = 0;
	HD_CHECK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, g.stream));
	HD_CHECK(hipStreamSynchronize(g.stream));
	if (g.h_out.reserve(total + 16))
		return HD_E_NOMEM;
	HD_CHECK(hipMemcpyAsync(g.h_out.p, g.d_packed.p, total, hipMemcpyDeviceToHost, g.stream));
	HD_CHECK(hipStreamSynchronize(g.stream));
	{
		// our stream is drained: launches on other streams need not wait for its events
		std::lock_guard<std::mutex> lk2(g.mu_dev);
		if (g.st_tiles == g.stream)
			g.tiles_used = false;
		if (g.st_tok == g.stream)
			g.tok_used = false;
	}
	const uint8_t *hp = (const uint8_t *)g.h_out.p;
	const uint32_t *h_crc = h_olen + nblocks;
	const int32_t *h_st = (const int32_t *)(h_crc + nblocks);
	size_t po = 0;
	for (uint32_t i = 0; i < nblocks; i++) {
		out_len[i] = h_olen[i];
		if (crc32)
			crc32[i] = h_crc[i];
		if (status)
			status[i] = h_st[i];
		memcpy(out + (uint64_t)i * out_stride, hp + po, h_olen[i]);
		po += h_olen[i];
	}
	return 0;
}

static int batch_inflate_host(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks,
			      uint8_t *out, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
			      uint32_t *crc32, int32_t *status, uint32_t flags)
{
	Ctx &g = cur();
	int r = ensure();
	if (r)
		return r;
	if (nblocks == 0)
		return 0;
	if (!in_off || !in_len || !out_off || !out_cap || !out_len)
		return HD_E_ARG;
	std::lock_guard<std::mutex> lk(g.mu);
	if ((r = bind_device()))
		return r;
	size_t in_total = 0, out_total = 0;
	for (uint32_t i = 0; i < nblocks; i++) {
		if (in_len[i] >= HD_INFLATE_MAX_IN)
			return HD_E_ARG;
		in_total += up16(in_len[i]);
		out_total += up16(out_cap[i]);
	}
	const size_t meta_bytes = (size_t)nblocks * (8 + 4 + 8 + 4 + 4 + 4 + 4);
	if (g.h_in.reserve(in_total + 16) || g.h_meta.reserve(meta_bytes) || g.d_in.reserve(in_total + 16) ||
	    g.d_meta.reserve(meta_bytes) || g.d_slots.reserve(out_total + 16) || g.h_out.reserve(out_total + 16))
		return HD_E_NOMEM;
	uint8_t *hin = (uint8_t *)g.h_in.p;
	uint64_t *h_ioff = (uint64_t *)g.h_meta.p;
	uint64_t *h_ooff = h_ioff + nblocks;
	uint32_t *h_ilen = (uint32_t *)(h_ooff + nblocks);
	uint32_t *h_ocap = h_ilen + nblocks;
	size_t io = 0, oo = 0;
	for (uint32_t i = 0; i < nblocks; i++) {
		h_ioff[i] = io;
		h_ooff[i] = oo;
		h_ilen[i] = in_len[i];
		h_ocap[i] = out_cap[i];
		if (in_len[i])
			memcpy(hin + io, in + in_off[i], in_len[i]);
		io += up16(in_len[i]);
		oo += up16(out_cap[i]);
	}
	uint8_t *dm = (uint8_t *)g.d_meta.p;
	uint64_t *d_ioff = (uint64_t *)dm;
	uint64_t *d_ooff = d_ioff + nblocks;
	uint32_t *d_ilen = (uint32_t *)(d_ooff + nblocks);
	uint32_t *d_ocap = d_ilen + nblocks;
	uint32_t *d_olen = d_ocap + nblocks;
	uint32_t *d_crc = d_olen + nblocks;
	int32_t *d_st = (int32_t *)(d_crc + nblocks);
	HD_CHECK(hipMemcpyAsync(g.d_in.p, hin, in_total, hipMemcpyHostToDevice, g.stream));
	HD_CHECK(hipMemcpyAsync(dm, g.h_meta.p, (size_t)nblocks * 24, hipMemcpyHostToDevice, g.stream));
	r = batch_inflate_dev(g.d_in.p, d_ioff, d_ilen, nblocks, g.d_slots.p, d_ooff, d_ocap, d_olen, crc32 ? d_crc : nullptr,
			      d_st, g.stream, flags);
	if (r)
		return r;
	uint32_t *h_olen = (uint32_t *)g.h_meta.p + 6 * (size_t)nblocks;     // past the 24 B/blk inputs
	HD_CHECK(hipMemcpyAsync(h_olen, d_olen, (size_t)nblocks * 12, hipMemcpyDeviceToHost, g.stream));
	HD_CHECK(hipMemcpyAsync(g.h_out.p, g.d_slots.p, out_total, hipMemcpyDeviceToHost, g.stream));
	HD_CHECK(hipStreamSynchronize(g.stream));
	const uint32_t *h_crc = h_olen + nblocks;
	const int32_t *h_st = (const int32_t *)(h_crc + nblocks);
	for (uint32_t i = 0; i < nblocks; i++) {
		out_len[i] = h_olen[i];
		if (crc32)
			crc32[i] = h_crc[i];
		if (status)
			status[i] = h_st[i];
		if (h_st[i] == 0 && h_olen[i])
			memcpy(out + out_off[i], (const uint8_t *)g.h_out.p + h_ooff[i], h_olen[i]);
	}
	return 0;
}

int hipdeflate_batch_inflate(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks,
			     uint8_t *out, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
			     uint32_t *crc32, int32_t *status)
{
	return batch_inflate_host(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, crc32, status, 0);
}

int hipdeflate_batch_inflate_flush(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len, uint32_t nblocks,
				   uint8_t *out, const uint64_t *out_off, const uint32_t *out_cap, uint32_t *out_len,
				   uint32_t *crc32, int32_t *status)
{
	return batch_inflate_host(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, crc32, status, hd::INF_FLUSHED);
}

/* ---- streaming encoder ---------------------------------------------------------- */

struct PipeSlot {
	Buf h_in{ nullptr, 0, true }, h_out{ nullptr, 0, true }, h_meta{ nullptr, 0, true };
	std::vector<uint64_t> doff;      // member offsets of the held result, 8-byte aligned (pipe_members)
	Buf d_in, d_meta, d_slots, d_packed;
	hipStream_t st = nullptr;
	size_t nbytes = 0;
	uint32_t nb = 0;
	int state = 0;               // 0 free, 1 being filled, 2 submitted, 3 result held by the caller
};

struct hipdeflate_pipe {
	int ctx = 0;                                 // entry of the device list the pipe lives on
	int level, frame, depth;
	uint32_t block, per_batch;
	size_t slot_stride;
	std::vector<PipeSlot> slots;
	std::mutex mu;
	std::condition_variable cv;
	uint64_t n_in = 0, n_sub = 0, n_out = 0;     // slots handed out / submitted / fetched
	int held = -1;
};

hipdeflate_pipe *hipdeflate_pipe_open(int level, int frame, uint32_t block_bytes, uint32_t blocks_per_batch, int depth)
{
	if (ensure() || bind_device())
		return nullptr;
	if (frame < HD_FRAME_RAW || frame > HD_FRAME_GZIP || !block_bytes || (block_bytes & 15) || !blocks_per_batch ||
	    depth < 2 || depth > 16 || (uint64_t)block_bytes * blocks_per_batch > 0xffff0000ull)
		return nullptr;
	hipdeflate_pipe *p = new hipdeflate_pipe;
	p->ctx = t_cur;
	p->level = level;
	p->frame = frame;
	p->depth = depth;
	p->block = block_bytes;
	p->per_batch = blocks_per_batch;
	// a member never needs more than the stored form + the largest frame
	p->slot_stride = (size_t)hipdeflate_bound(block_bytes, level);
	if (frame == HD_FRAME_BGZF && p->slot_stride > 65536)
		p->slot_stride = 65536;
	p->slots.resize(depth);
	// A slot's memory is pinned / allocated when the slot is first handed out (pipe_slot_alloc): pinning costs ~0.3 ms per
	// MiB, and a stream that ends after one batch -- or a caller that wants its first bytes soon -- should not wait for
	// `depth` batches' worth of it.  Only the streams are made here.
	for (PipeSlot &s : p->slots) {
		if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) {
			hipdeflate_pipe_close(p);
			return nullptr;
		}
	}
	return p;
}

// (the caller is the one thread that does input() / submit() on this pipe)
static int pipe_slot_alloc(hipdeflate_pipe *p, PipeSlot &s)
{
	if (s.h_in.p)
		return 0;
	const size_t in_cap = (size_t)p->block * p->per_batch;
	const size_t meta = (size_t)p->per_batch * (8 + 4 + 4 + 4 + 4 + 8) + 16;
	if (s.h_in.reserve(in_cap) || s.d_in.reserve(in_cap + 16) || s.h_meta.reserve(meta) || s.d_meta.reserve(meta) ||
	    s.d_slots.reserve(p->slot_stride * p->per_batch + 16) || s.d_packed.reserve(p->slot_stride * p->per_batch + 16))
		return HD_E_NOMEM;
	return 0;
}

hipdeflate_pipe *hipdeflate_pipe_open_on(int index, int level, int frame, uint32_t block_bytes, uint32_t blocks_per_batch,
					 int depth)
{
	if (index < 0 || index >= hipdeflate_device_count())
		return nullptr;
	const OnCtx on(index);
	return hipdeflate_pipe_open(level, frame, block_bytes, blocks_per_batch, depth);
}

uint8_t *hipdeflate_pipe_input(hipdeflate_pipe *p, size_t *cap)
{
	if (!p)
		return nullptr;
	std::unique_lock<std::mutex> lk(p->mu);
	PipeSlot &s = p->slots[p->n_in % p->depth];
	if (s.state == 1)
		return nullptr;                          // input() twice without submit()
	p->cv.wait(lk, [&] { return s.state == 0; });
	lk.unlock();
	{
		const OnCtx on(p->ctx);
		if (ensure() || bind_device() || pipe_slot_alloc(p, s))
			return nullptr;
	}
	lk.lock();
	s.state = 1;
	if (cap)
		*cap = (size_t)p->block * p->per_batch;
	return (uint8_t *)s.h_in.p;
}

int hipdeflate_pipe_submit(hipdeflate_pipe *p, size_t nbytes)
{
	if (!p)
		return HD_E_ARG;
	PipeSlot *sp;
	{
		std::lock_guard<std::mutex> lk(p->mu);
		sp = &p->slots[p->n_in % p->depth];
		if (sp->state != 1 || nbytes > (size_t)p->block * p->per_batch)
			return HD_E_ARG;
	}
	PipeSlot &s = *sp;
	const OnCtx on(p->ctx);
	int r = ensure();
	if (r || (r = bind_device()))
		return r;
	s.nbytes = nbytes;
	s.nb = (uint32_t)((nbytes + p->block - 1) / p->block);
	if (s.nb) {
		// block table: offsets i * block (16-byte aligned), the last block may be short
		uint64_t *h_off = (uint64_t *)s.h_meta.p;
		uint32_t *h_len = (uint32_t *)(h_off + s.nb);
		for (uint32_t i = 0; i < s.nb; i++) {
			h_off[i] = (uint64_t)i * p->block;
			h_len[i] = i + 1 < s.nb ? p->block : (uint32_t)(nbytes - (size_t)i * p->block);
		}
		uint8_t *dm = (uint8_t *)s.d_meta.p;
		uint64_t *d_off = (uint64_t *)dm;
		uint32_t *d_len = (uint32_t *)(d_off + s.nb);
		uint32_t *d_olen = d_len + s.nb;
		uint32_t *d_crc = d_olen + s.nb;
		int32_t *d_st = (int32_t *)(d_crc + s.nb);
		uint64_t *d_doff = (uint64_t *)(((uintptr_t)(d_st + s.nb) + 7) & ~(uintptr_t)7);
		uint64_t *d_total = d_doff + s.nb;
		HD_CHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, nbytes, hipMemcpyHostToDevice, s.st));
		HD_CHECK(hipMemcpyAsync(d_off, h_off, (size_t)s.nb * 12, hipMemcpyHostToDevice, s.st));
		if ((r = batch_deflate_dev_impl(s.d_in.p, d_off, d_len, s.nb, p->level, p->frame, s.d_slots.p,
						p->slot_stride, (uint32_t)p->slot_stride, d_olen, d_crc, d_st, s.st, p->block)))
			return r;
		if ((r = hipdeflate_scan_sizes_dev(d_olen, s.nb, 0, d_doff, d_total, s.st)))
			return r;
		if ((r = hipdeflate_compact_dev(s.d_slots.p, p->slot_stride, d_olen, d_doff, s.nb, s.d_packed.p, s.st)))
			return r;
		// sizes, status and the total come back first; the payload follows in result()
		HD_CHECK(hipMemcpyAsync(s.h_meta.p, d_olen, (size_t)((uint8_t *)(d_total + 1) - (uint8_t *)d_olen),
					hipMemcpyDeviceToHost, s.st));
	}
	std::lock_guard<std::mutex> lk(p->mu);
	s.state = 2;
	p->n_in++;
	p->n_sub++;
	p->cv.notify_all();
	return 0;
}

int hipdeflate_pipe_result(hipdeflate_pipe *p, const uint8_t **data, size_t *nbytes, uint32_t *nblocks)
{
	if (!p || !data || !nbytes)
		return HD_E_ARG;
	PipeSlot *sp;
	{
		std::unique_lock<std::mutex> lk(p->mu);
		if (p->held >= 0) {                      // the previous result goes back to the pool
			p->slots[p->held].state = 0;
			p->held = -1;
			p->cv.notify_all();
		}
		if (p->n_out == p->n_sub)
			return HD_E_ARG;                 // nothing pending
		sp = &p->slots[p->n_out % p->depth];
	}
	PipeSlot &s = *sp;
	const OnCtx on(p->ctx);
	int r = ensure();
	if (r || (r = bind_device()))
		return r;
	int bad = 0;
	size_t total = 0;
	if (s.nb) {
		HD_CHECK(hipStreamSynchronize(s.st));
		// the device layout from d_olen on: olen[nb], crc[nb], status[nb], doff[nb] (u64), total (u64);
		// d_meta is 256-byte aligned and 24 nb bytes precede doff, so there is no padding
		const uint32_t *h_olen = (const uint32_t *)s.h_meta.p;
		const int32_t *h_st = (const int32_t *)(h_olen + 2 * s.nb);
		uint64_t t64;
		memcpy(&t64, (const uint8_t *)s.h_meta.p + (size_t)20 * s.nb, 8);
		total = (size_t)t64;
		// (doff[] sits 12 nb bytes into the pinned table: only 4-byte aligned for an odd nb)
		s.doff.resize(s.nb);
		memcpy(s.doff.data(), (const uint8_t *)s.h_meta.p + (size_t)12 * s.nb, (size_t)8 * s.nb);
		for (uint32_t i = 0; i < s.nb; i++)
			bad |= h_st[i] != 0;
		// the pinned landing buffer follows what the data needs (+ 25 %), not the worst case of the slots
		if (s.h_out.reserve(total + 16))
			return HD_E_NOMEM;
		HD_CHECK(hipMemcpyAsync(s.h_out.p, s.d_packed.p, total, hipMemcpyDeviceToHost, s.st));
		HD_CHECK(hipStreamSynchronize(s.st));
	}
	*data = (const uint8_t *)s.h_out.p;
	*nbytes = total;
	if (nblocks)
		*nblocks = s.nb;
	std::lock_guard<std::mutex> lk(p->mu);
	s.state = 3;
	p->held = (int)(p->n_out % p->depth);
	p->n_out++;
	return bad ? 1 : 0;
}

int hipdeflate_pipe_members(hipdeflate_pipe *p, const uint32_t **out_len, const uint64_t **dst_off, const uint32_t **crc32)
{
	if (!p)
		return HD_E_ARG;
	std::lock_guard<std::mutex> lk(p->mu);
	if (p->held < 0)
		return HD_E_ARG;                         // no result is held
	const PipeSlot &s = p->slots[p->held];
	// the layout pipe_submit() copied back: olen[nb], crc[nb], status[nb], doff[nb] (u64), total (u64)
	const uint32_t *h_olen = (const uint32_t *)s.h_meta.p;
	const bool none = s.nb == 0;             // an empty batch: the pinned table holds an earlier batch's figures
	if (out_len)
		*out_len = none ? nullptr : h_olen;
	if (crc32)
		*crc32 = none ? nullptr : h_olen + s.nb;
	if (dst_off)
		*dst_off = none ? nullptr : s.doff.data();
	return 0;
}

void hipdeflate_pipe_close(hipdeflate_pipe *p)
{
	if (!p)
		return;
	const OnCtx on(p->ctx);
	Ctx &g = cur();
	(void)hipSetDevice(g.device);
	(void)hipDeviceSynchronize();
	for (PipeSlot &s : p->slots) {
		s.h_in.release();
		s.h_out.release();
		s.h_meta.release();
		s.d_in.release();
		s.d_meta.release();
		s.d_slots.release();
		s.d_packed.release();
		if (s.st) {
			// everything on the stream has finished: nobody needs to wait for its events any more
			std::lock_guard<std::mutex> lk(g.mu_dev);
			if (g.st_tiles == s.st)
				g.tiles_used = false;
			if (g.st_tok == s.st)
				g.tok_used = false;
			(void)hipStreamDestroy(s.st);
		}
	}
	delete p;
}

/* ---- streaming decoder ---------------------------------------------------------- */

struct hipdeflate_unpipe {
	int ctx = 0;
	int depth;
	uint32_t max_members;
	size_t in_cap, out_cap;
	std::vector<PipeSlot> slots;
	std::mutex mu;
	std::condition_variable cv;
	uint64_t n_in = 0, n_sub = 0, n_out = 0;
	int held = -1;
};

hipdeflate_unpipe *hipdeflate_unpipe_open(uint32_t max_members, size_t in_cap, size_t out_cap, int depth)
{
	if (ensure() || bind_device())
		return nullptr;
	if (!max_members || !in_cap || !out_cap || depth < 2 || depth > 16 || in_cap > 0xffff0000ull)
		return nullptr;
	hipdeflate_unpipe *p = new hipdeflate_unpipe;
	p->ctx = t_cur;
	p->depth = depth;
	p->max_members = max_members;
	p->in_cap = in_cap;
	p->out_cap = out_cap;
	p->slots.resize(depth);
	for (PipeSlot &s : p->slots) {                            // memory: when a slot is first handed out (unpipe_slot_alloc)
		if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) {
			hipdeflate_unpipe_close(p);
			return nullptr;
		}
	}
	return p;
}

static int unpipe_slot_alloc(hipdeflate_unpipe *p, PipeSlot &s)
{
	if (s.h_in.p)
		return 0;
	const size_t meta = (size_t)p->max_members * (8 + 8 + 4 + 4 + 4 + 4 + 4) + 64;
	if (s.h_in.reserve(p->in_cap) || s.d_in.reserve(p->in_cap + 16) || s.h_meta.reserve(meta) || s.d_meta.reserve(meta) ||
	    s.d_slots.reserve(p->out_cap + 16) || s.h_out.reserve(p->out_cap + 16))
		return HD_E_NOMEM;
	return 0;
}

hipdeflate_unpipe *hipdeflate_unpipe_open_on(int index, uint32_t max_members, size_t in_cap, size_t out_cap, int depth)
{
	if (index < 0 || index >= hipdeflate_device_count())
		return nullptr;
	const OnCtx on(index);
	return hipdeflate_unpipe_open(max_members, in_cap, out_cap, depth);
}

uint8_t *hipdeflate_unpipe_input(hipdeflate_unpipe *p, size_t *cap)
{
	if (!p)
		return nullptr;
	std::unique_lock<std::mutex> lk(p->mu);
	PipeSlot &s = p->slots[p->n_in % p->depth];
	if (s.state == 1)
		return nullptr;
	p->cv.wait(lk, [&] { return s.state == 0; });
	lk.unlock();
	{
		const OnCtx on(p->ctx);
		if (ensure() || bind_device() || unpipe_slot_alloc(p, s))
			return nullptr;
	}
	lk.lock();
	s.state = 1;
	if (cap)
		*cap = p->in_cap;
	return (uint8_t *)s.h_in.p;
}

int hipdeflate_unpipe_submit(hipdeflate_unpipe *p, const uint64_t *in_off, const uint32_t *in_len,
			     const uint32_t *out_size, uint32_t nmembers)
{
	if (!p || nmembers > p->max_members || (nmembers && (!in_off || !in_len || !out_size)))
		return HD_E_ARG;
	PipeSlot *sp;
	{
		std::lock_guard<std::mutex> lk(p->mu);
		sp = &p->slots[p->n_in % p->depth];
		if (sp->state != 1)
			return HD_E_ARG;
	}
	PipeSlot &s = *sp;
	const OnCtx on(p->ctx);
	int r = ensure();
	if (r || (r = bind_device()))
		return r;
	// pinned table: ioff[n] ooff[n] (u64) | ilen[n] ocap[n] (u32); results olen[n] crc[n] st[n] behind it
	const uint32_t n = nmembers;
	uint64_t *h_ioff = (uint64_t *)s.h_meta.p, *h_ooff = h_ioff + n;
	uint32_t *h_ilen = (uint32_t *)(h_ooff + n), *h_ocap = h_ilen + n;
	size_t in_end = 0, osum = 0;
	for (uint32_t i = 0; i < n; i++) {
		if (in_len[i] >= HD_INFLATE_MAX_IN)
			return HD_E_ARG;
		h_ioff[i] = in_off[i];
		h_ilen[i] = in_len[i];
		h_ooff[i] = osum;
		h_ocap[i] = out_size[i];
		osum += out_size[i];
		if (in_off[i] + in_len[i] > in_end)
			in_end = (size_t)(in_off[i] + in_len[i]);
	}
	if (in_end > p->in_cap || osum > p->out_cap)
		return HD_E_ARG;
	s.nb = n;
	s.nbytes = osum;
	if (n) {
		uint8_t *dm = (uint8_t *)s.d_meta.p;
		uint64_t *d_ioff = (uint64_t *)dm, *d_ooff = d_ioff + n;
		uint32_t *d_ilen = (uint32_t *)(d_ooff + n), *d_ocap = d_ilen + n, *d_olen = d_ocap + n;
		int32_t *d_st = (int32_t *)(d_olen + n);
		HD_CHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_end, hipMemcpyHostToDevice, s.st));
		HD_CHECK(hipMemcpyAsync(dm, s.h_meta.p, (size_t)n * 24, hipMemcpyHostToDevice, s.st));
		if ((r = hipdeflate_batch_inflate_dev(s.d_in.p, d_ioff, d_ilen, n, s.d_slots.p, d_ooff, d_ocap, d_olen, nullptr,
						      d_st, s.st)))
			return r;
		HD_CHECK(hipMemcpyAsync((uint8_t *)s.h_meta.p + (size_t)n * 24, d_olen, (size_t)n * 8, hipMemcpyDeviceToHost,
					s.st));
		if (osum)
			HD_CHECK(hipMemcpyAsync(s.h_out.p, s.d_slots.p, osum, hipMemcpyDeviceToHost, s.st));
	}
	std::lock_guard<std::mutex> lk(p->mu);
	s.state = 2;
	p->n_in++;
	p->n_sub++;
	p->cv.notify_all();
	return 0;
}

int hipdeflate_unpipe_result(hipdeflate_unpipe *p, const uint8_t **data, size_t *nbytes)
{
	if (!p || !data || !nbytes)
		return HD_E_ARG;
	PipeSlot *sp;
	{
		std::unique_lock<std::mutex> lk(p->mu);
		if (p->held >= 0) {
			p->slots[p->held].state = 0;
			p->held = -1;
			p->cv.notify_all();
		}
		if (p->n_out == p->n_sub)
			return HD_E_ARG;
		sp = &p->slots[p->n_out % p->depth];
	}
	PipeSlot &s = *sp;
	const OnCtx on(p->ctx);
	int r = ensure();
	if (r || (r = bind_device()))
		return r;
	int verdict = 0;
	if (s.nb) {
		HD_CHECK(hipStreamSynchronize(s.st));
		const uint32_t n = s.nb;
		const uint32_t *h_ocap = (const uint32_t *)((const uint8_t *)s.h_meta.p + (size_t)n * 20);
		const uint32_t *h_olen = h_ocap + n;
		const int32_t *h_st = (const int32_t *)(h_olen + n);
		for (uint32_t i = 0; i < n && !verdict; i++)
			verdict = h_st[i] ? h_st[i] : (h_olen[i] != h_ocap[i] ? HD_INSUFFICIENT_SPACE : 0);
	}
	*data = (const uint8_t *)s.h_out.p;
	*nbytes = s.nbytes;
	std::lock_guard<std::mutex> lk(p->mu);
	s.state = 3;
	p->held = (int)(p->n_out % p->depth);
	p->n_out++;
	return verdict;
}

void hipdeflate_unpipe_close(hipdeflate_unpipe *p)
{
	if (!p)
		return;
	const OnCtx on(p->ctx);
	Ctx &g = cur();
	(void)hipSetDevice(g.device);
	(void)hipDeviceSynchronize();
	for (PipeSlot &s : p->slots) {
		s.h_in.release();
		s.h_out.release();
		s.h_meta.release();
		s.d_in.release();
		s.d_meta.release();
		s.d_slots.release();
		if (s.st) {
			// everything on the stream has finished: nobody needs to wait for its events any more
			std::lock_guard<std::mutex> lk(g.mu_dev);
			if (g.st_tiles == s.st)
				g.tiles_used = false;
			if (g.st_tok == s.st)
				g.tok_used = false;
			(void)hipStreamDestroy(s.st);
		}
	}
	delete p;
}

/* ---- latency contexts: small synchronous batches ------------------------------------
 * Everything a call needs is allocated once: pinned, device-visible memory for the blocks, the members and the
 * tables (the kernels read the input and write the members over PCIe themselves -- no copy engine, no staging
 * copy under a lock), device scratch for the segments, one stream.  run() = a handful of launches + one
 * synchronisation.  Contexts are independent: two of them overlap on the device. */
struct hipdeflate_lat {
	int ctx = 0;                                                    // entry of the device list
	int level = 1, frame = HD_FRAME_BGZF;
	uint32_t max_blocks = 0, in_stride = 0, slot = 0;
	hipStream_t st = nullptr;
	Buf h_in{ nullptr, 0, true }, h_out{ nullptr, 0, true }, h_meta{ nullptr, 0, true };
	Buf d_scratch;
	uint8_t *din = nullptr, *dout = nullptr, *dmeta = nullptr;      // device views of the pinned buffers
	bool latency = true;
	uint32_t seg = 0, seg_limit = 0, S = 0;                         // segment bytes, limit, segment slots per block
	size_t meta_seg = 0;                                            // offset of the segment table in the meta buffer
	// completion without hipStreamSynchronize (HIPDEFLATE_LAT_POLL): the run's last kernel stores the run's number here
	size_t flag_off = 0;                                            // ... in the meta buffer (pinned, device-visible)
	Buf d_count;                                                    // the word its workgroups count themselves off on
	uint32_t epoch = 0;
};

hipdeflate_lat *hipdeflate_lat_open(int level, int frame, uint32_t max_blocks, uint32_t max_block_bytes)
{
	if (ensure() || bind_device())
		return nullptr;
	const int fr = frame & ~HD_FRAME_LATENCY;
	if (fr < HD_FRAME_RAW || fr > HD_FRAME_GZIP || !max_blocks || max_blocks > 65536 || !max_block_bytes ||
	    max_block_bytes > (64u << 20))
		return nullptr;
	hipdeflate_lat *c = new hipdeflate_lat;
	c->ctx = t_cur;
	c->level = level;
	c->frame = fr;
	c->latency = (frame & HD_FRAME_LATENCY) != 0;
	c->max_blocks = max_blocks;
	c->in_stride = (uint32_t)up16(max_block_bytes);
	c->slot = (uint32_t)hipdeflate_bound(max_block_bytes, level);
	if (fr == HD_FRAME_BGZF && c->slot > 65536)
		c->slot = 65536;
	const uint32_t seg_lim = c->latency ? HD_LAT_SEG_BYTES(level) : HD_SEG_LIMIT;
	c->seg = c->latency ? HD_LAT_SEG_BYTES(level) : HD_SEG_BYTES;
	c->seg_limit = (level >= 1 && level < HD_WG_LEVEL && c->slot > seg_lim) ? seg_lim : 0;
	c->S = c->seg_limit ? hd::seg_slots_per_block(c->slot, c->seg) : 0;
	// [ in_off u64 | in_len, out_len, crc, status u32 | seg_off u64 [max_blocks * S] | seg_len u32 [max_blocks * S] ]
	c->meta_seg = (((size_t)max_blocks * (8 + 4 + 4 + 4 + 4)) + 15) & ~(size_t)15;
	const size_t meta = c->meta_seg + (size_t)max_blocks * c->S * 12 + 64;
	c->flag_off = meta - 16;
	// (the workgroup levels' records are sized by the longest block, not by the slot)
	const uint64_t scr = scratch_need(max_blocks, level >= HD_WG_LEVEL ? c->in_stride : c->slot, level, c->latency);
	if (c->h_in.reserve((size_t)max_blocks * c->in_stride) || c->h_out.reserve((size_t)max_blocks * c->slot) ||
	    c->h_meta.reserve(meta) || (scr && c->d_scratch.reserve(scr)) || c->d_count.reserve(16) ||
	    hipMemset(c->d_count.p, 0, 16) != hipSuccess ||
	    hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&c->din, c->h_in.p, 0) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&c->dout, c->h_out.p, 0) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&c->dmeta, c->h_meta.p, 0) != hipSuccess) {
		hipdeflate_lat_close(c);
		return nullptr;
	}
	memset(c->h_meta.p, 0, meta);
	uint64_t *off = (uint64_t *)c->h_meta.p;
	for (uint32_t i = 0; i < max_blocks; i++)
		off[i] = (uint64_t)i * c->in_stride;
	return c;
}

hipdeflate_lat *hipdeflate_lat_open_on(int index, int level, int frame, uint32_t max_blocks, uint32_t max_block_bytes)
{
	if (index < 0 || index >= hipdeflate_device_count())
		return nullptr;
	const OnCtx on(index);
	return hipdeflate_lat_open(level, frame, max_blocks, max_block_bytes);
}

uint8_t *hipdeflate_lat_input(hipdeflate_lat *c, uint32_t i)
{
	return c && i < c->max_blocks ? (uint8_t *)c->h_in.p + (size_t)i * c->in_stride : nullptr;
}

// the run with the frame, the mode and the room per member chosen by the call (the per-block codecs: a context per
// thread serves hip_deflate and hip_deflate_flush, and falls back to the ordinary form when the caller's room is
// below the latency form's worst case)
static int lat_run_ex(hipdeflate_lat *c, const uint32_t *in_len, uint32_t n, int frame, bool latency, uint32_t out_cap)
{
	if (!c || n > c->max_blocks || (n && !in_len))
		return HD_E_ARG;
	if (!n)
		return 0;
	// (a context may outlive hipdeflate_shutdown(), which frees the shared tables: ensure() brings them back)
	const OnCtx on(c->ctx);
	Ctx &g = cur();
	int r = ensure();
	if (r || (r = bind_device()))
		return r;
	const uint32_t mb = c->max_blocks;
	const uint32_t seg_limit = latency ? c->seg_limit : 0;           // (a context's slots are far below HD_SEG_LIMIT)
	uint32_t *h_len = (uint32_t *)((uint64_t *)c->h_meta.p + mb);
	uint32_t hint = hd::HD_HINT_NO_WHOLE | hd::HD_HINT_NO_SEG;
	uint64_t *h_soff = (uint64_t *)((uint8_t *)c->h_meta.p + c->meta_seg);
	uint32_t *h_slen = (uint32_t *)(h_soff + (size_t)mb * c->S);
	for (uint32_t i = 0; i < n; i++) {
		if (in_len[i] > c->in_stride)
			return HD_E_ARG;
		h_len[i] = in_len[i];
		const bool segd = seg_limit && in_len[i] > seg_limit;
		hint &= segd ? ~hd::HD_HINT_NO_SEG : ~hd::HD_HINT_NO_WHOLE;
		// the segment table k_seg_table would make (hd_segment.hpp), straight into device-visible memory
		for (uint32_t k = 0; k < c->S; k++) {
			const uint64_t o = (uint64_t)k * c->seg;
			const uint32_t sl = (segd && o < in_len[i]) ? (in_len[i] - o < c->seg ? (uint32_t)(in_len[i] - o) : c->seg) : 0u;
			h_soff[(size_t)i * c->S + k] = (uint64_t)i * c->in_stride + (sl ? o : 0);
			h_slen[(size_t)i * c->S + k] = sl;
		}
	}
	uint64_t *d_off = (uint64_t *)c->dmeta;
	uint32_t *d_len = (uint32_t *)(d_off + mb), *d_olen = d_len + mb, *d_crc = d_olen + mb;
	int32_t *d_st = (int32_t *)(d_crc + mb);
	hd::DeflateArgs a;
	a.in = c->din;
	a.in_off = d_off;
	a.in_len = d_len;
	a.nblocks = n;
	a.frame = frame;
	a.level = c->level;
	a.out = c->dout;
	a.out_stride = c->slot;
	a.out_cap = out_cap < c->slot ? out_cap : c->slot;
	a.out_len = d_olen;
	a.crc = d_crc;
	a.status = d_st;
	a.ct = g.d_ct;
	a.scratch = (uint8_t *)c->d_scratch.p;
	a.first = 0;
	a.count = 0;
	a.skip_small = 0;
	a.split_max = c->level >= HD_WG_LEVEL ? c->in_stride : hd::split_max_block(c->slot, c->slot);
	a.split_ovf = nullptr;
	a.seg_bytes = c->seg;
	a.seg_limit = seg_limit;
	// the workgroup levels: the member written by a workgroup (hd_emit_wg.hpp) when no block can be longer than 64 KiB
	a.lat = (latency && c->level >= HD_WG_LEVEL && a.split_max <= hd::EW_BLOCK_MAX) ? 1u : 0u;
	a.stalls = g.d_stalls;
	a.hint = seg_limit ? hint : 0;
	a.host_seg_off = seg_limit ? (const uint64_t *)(c->dmeta + c->meta_seg) : nullptr;
	a.host_seg_len = seg_limit ? (const uint32_t *)((const uint64_t *)(c->dmeta + c->meta_seg) + (size_t)mb * c->S) : nullptr;
	// the segmented path's last kernel can report by itself (members of up to 64 segments, at least one segmented block)
	static const bool poll_on = [] { const char *e = getenv("HIPDEFLATE_LAT_POLL"); return e && *e && *e != '0'; }();
	const bool poll = poll_on && seg_limit && c->S <= 64 && !(a.hint & hd::HD_HINT_NO_SEG);
	volatile uint32_t *flag = (volatile uint32_t *)((uint8_t *)c->h_meta.p + c->flag_off);
	if (poll) {
		a.done_flag = (uint32_t *)(c->dmeta + c->flag_off);
		a.done_count = (uint32_t *)c->d_count.p;
		a.done_epoch = ++c->epoch ? c->epoch : ++c->epoch;      // (never 0: the word starts as 0)
	}
	if ((r = launch_deflate(a, c->level, c->st)))
		return r;
	if (poll) {
		// ~100 us of kernels: look at the word; should it not come (a fault), the stream's own wait tells
		const auto t0 = std::chrono::steady_clock::now();
		for (uint32_t spins = 0; __atomic_load_n((const uint32_t *)flag, __ATOMIC_ACQUIRE) != a.done_epoch; spins++) {
			__builtin_ia32_pause();
			if ((spins & 4095) == 4095 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
				HD_CHECK(hipStreamSynchronize(c->st));
				break;
			}
		}
		return 0;
	}
	HD_CHECK(hipStreamSynchronize(c->st));
	return 0;
}

int hipdeflate_lat_run(hipdeflate_lat *c, const uint32_t *in_len, uint32_t n)
{
	return c ? lat_run_ex(c, in_len, n, c->frame, c->latency, c->slot) : HD_E_ARG;
}

const uint8_t *hipdeflate_lat_output(hipdeflate_lat *c, uint32_t i, uint32_t *out_len, uint32_t *crc32, int32_t *status)
{
	if (!c || i >= c->max_blocks)
		return nullptr;
	const uint32_t mb = c->max_blocks;
	const uint32_t *h_olen = (const uint32_t *)((const uint64_t *)c->h_meta.p + mb) + mb;
	if (out_len)
		*out_len = h_olen[i];
	if (crc32)
		*crc32 = h_olen[mb + i];
	if (status)
		*status = (int32_t)h_olen[2 * mb + i];
	return (const uint8_t *)c->h_out.p + (size_t)i * c->slot;
}

void hipdeflate_lat_close(hipdeflate_lat *c)
{
	if (!c)
		return;
	if (g_all[c->ctx].ready)
		(void)hipSetDevice(g_all[c->ctx].device);
	if (c->st) {
		(void)hipStreamSynchronize(c->st);
		(void)hipStreamDestroy(c->st);
	}
	c->h_in.release();
	c->h_out.release();
	c->h_meta.release();
	c->d_scratch.release();
	c->d_count.release();
	delete c;
}

/* ---- per-block codecs (zlibutil_code_enc / zlibutil_code_dec) --------------- */

// One block per call from each of the caller's threads (zlibutil_buffer_code as a pthread start routine,
// applet/7bgzf.c:211 -- which creates a THREAD PER BLOCK): a call borrows a latency context (blocks up to 64 KiB) from a
// small pool keyed by level and gives it back, so a short-lived thread reuses the pinned buffers, scratch and stream of
// the one before it -- no hipHostMalloc / hipFree per block, no HIP call from a thread-exit destructor, no global
// lock around the coding itself: ~90 us per 0xff00-byte block and thread, all threads at once.  hipdeflate_shutdown()
// closes the pooled contexts.
constexpr uint32_t CODEC_BLOCK = 0x10000;

static hipdeflate_lat *codec_acquire(int level)
{
	const int k = level < 0 ? 0 : level >= CODEC_LEVELS ? CODEC_LEVELS - 1 : level;
	{
		std::lock_guard<std::mutex> lk(g_codec_mu);
		std::vector<hipdeflate_lat *> &f = g_codec_free[k];
		if (!f.empty()) {
			hipdeflate_lat *c = f.back();
			f.pop_back();
			return c;
		}
	}
	// a new context: the per-block codecs' contexts are spread over the device list
	static unsigned rr = 0;
	const int ndev = hipdeflate_device_count();
	const unsigned turn = __atomic_fetch_add(&rr, 1u, __ATOMIC_RELAXED);
	return hipdeflate_lat_open_on(ndev > 0 ? (int)(turn % (unsigned)ndev) : 0, level, HD_FRAME_RAW | HD_FRAME_LATENCY, 1, CODEC_BLOCK);
}

static void codec_release(hipdeflate_lat *c, int level)
{
	const int k = level < 0 ? 0 : level >= CODEC_LEVELS ? CODEC_LEVELS - 1 : level;
	{
		std::lock_guard<std::mutex> lk(g_codec_mu);
		if (g_codec_free[k].size() < CODEC_POOL_MAX) {
			g_codec_free[k].push_back(c);
			return;
		}
	}
	hipdeflate_lat_close(c);
}

extern "C" int hd_codec_batch(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t slen, int level, int flush);
static bool codec_batching()
{
	static const bool on = [] {
		const char *e = getenv("HIPDEFLATE_CODEC_BATCH");
		return !(e && *e == '0');
	}();
	return on;
}

static int deflate_one(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level,
		       int frame)
{
	if (!dest || !destLen || (!source && sourceLen) || sourceLen > 0xffffffffu - 65536u)
		return HD_E_ARG;
	// Concurrent callers -- the reference creates a thread per block, -@N at once (applet/7bgzf.c:211) -- share launches: the
	// hook's micro-batcher with an engine per (level, frame) (bgzf_hook.c hd_codec_batch).  For a block whose room covers
	// the latency form's worst case AND the stored form the bytes do not depend on the room (what the reference's 1.5 x
	// allocation always gives), so callers with different rooms can sit in one batch; everything else takes its own context
	// below.  HIPDEFLATE_CODEC_BATCH=0 turns it off (A/B).
	if (sourceLen && sourceLen <= CODEC_BLOCK && level >= 0 && level <= 9 && codec_batching() && ensure() == 0) {
		const bool fl = frame == HD_FRAME_RAW_FLUSH;
		const uint32_t lat = HD_LAT_SEG_BYTES(level);
		const uint64_t need_lat = level >= 1 && level < HD_WG_LEVEL && sourceLen > lat ? HD_SEGN_WORST((uint64_t)sourceLen, lat, fl) : 0;
		const uint64_t need_st = HD_STORED_SIZE((uint64_t)sourceLen) + (fl ? 5u : 0u) + 8u;
		if (*destLen >= need_lat && *destLen >= need_st) {
			const int r = hd_codec_batch(dest, destLen, source, sourceLen, level, fl ? 1 : 0);
			if (r == 0 || r == 1)
				return r;                                /* (-1 / -2: not for a batch, or its engine is down) */
		}
	}
	if (sourceLen <= CODEC_BLOCK && ensure() == 0) {
		if (hipdeflate_lat *c = codec_acquire(level)) {
			const size_t cap = *destLen > c->slot ? c->slot : *destLen;
			const uint32_t lat = HD_LAT_SEG_BYTES(level);
			// latency form only when the room covers its worst case AND the context has a slot for every segment
			// (the workgroup levels: one stream in either mode, "latency" only picks the kernels that write it)
			const bool latency = level >= HD_WG_LEVEL ||
					     (level >= 1 && sourceLen > lat &&
					      cap >= HD_SEGN_WORST((uint64_t)sourceLen, lat, frame == HD_FRAME_RAW_FLUSH) &&
					      HD_SEGN_COUNT((uint32_t)sourceLen, lat) <= c->S);
			if (sourceLen)
				memcpy(hipdeflate_lat_input(c, 0), source, sourceLen);
			const uint32_t len = (uint32_t)sourceLen;
			int r = lat_run_ex(c, &len, 1, frame, latency, (uint32_t)cap);
			uint32_t olen = 0;
			int32_t st = 0;
			if (!r) {
				const uint8_t *m = hipdeflate_lat_output(c, 0, &olen, nullptr, &st);
				if (st || olen > *destLen)
					r = 1; /* !Z_OK, as libdeflate_deflate (lib/zlibutil.c:189) */
				else
					memcpy(dest, m, olen);
			}
			codec_release(c, level);
			if (!r)
				*destLen = olen;
			return r;
		}
	}
	uint64_t off = 0;
	uint32_t len = (uint32_t)sourceLen, olen = 0;
	int32_t st = 0;
	const size_t cap = *destLen > 0xfffffff0u ? 0xfffffff0u : *destLen;
	// One block per call: latency mode (several wavefronts for the block) whenever the room covers its worst case;
	// the ordinary form otherwise, so that -- as libdeflate_deflate -- the call succeeds whenever the stored form fits
	const uint32_t lat = HD_LAT_SEG_BYTES(level);
	if (level >= HD_WG_LEVEL || (level >= 1 && sourceLen > lat && cap >= HD_SEGN_WORST((uint64_t)sourceLen, lat, frame == HD_FRAME_RAW_FLUSH)))
		frame |= HD_FRAME_LATENCY;
	// the slot stride handed to the batch call only needs to cover `cap`
	int r = hipdeflate_batch_deflate(source, &off, &len, 1, level, frame, dest, up16(cap) ? up16(cap) : 16,
					 (uint32_t)cap, &olen, nullptr, &st);
	if (r)
		return r;
	if (st)
		return 1; /* !Z_OK, as libdeflate_deflate (lib/zlibutil.c:189) */
	*destLen = olen;
	return 0;
}

int hip_deflate(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level)
{
	return deflate_one(dest, destLen, source, sourceLen, level, HD_FRAME_RAW);
}

int hip_deflate_flush(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, int level)
{
	return deflate_one(dest, destLen, source, sourceLen, level, HD_FRAME_RAW_FLUSH);
}

/* ---- per-block decoder: hip_inflate / hip_inflate_flush -----------------------------------------------------------
 * The role of zlibutil_auto_inflate (lib/zlibutil.c:82-93 -> libdeflate_inflate :194-204) as the reference calls it: a
 * THREAD PER BLOCK, -@N of them at once (applet/7bgzf.c:330-345).  One stream is one wavefront's serial work, so what a
 * GPU can give such callers is all of them at once: concurrent calls are coalesced into one launch of the latency kernel
 * (k_inflate_lat: the whole window in LDS) over a batch's pinned, device-visible memory -- every caller copies its own
 * stream in and its own output out with no lock held, the kernel reads and writes that memory itself (no copy engine),
 * the first caller of a batch leads it (closes it once every caller that is inside and not already on the device has
 * joined, launches, waits), the others sleep on the batch's state word.  No process-wide lock around the decoding:
 * INFB_CTX batches can be collecting / running side by side, spread over the device list.  A stream or an output
 * larger than a batch's arena goes alone through the host-buffer batch call. */
constexpr uint32_t INFB_SLOTS = 64;                           // streams per batch
constexpr size_t INFB_IN_ARENA = (size_t)5 << 20;             // 64 x (64 KiB + the stored form's overhead) and room to spare
constexpr size_t INFB_OUT_ARENA = (size_t)4 << 20;            // 64 x 64 KiB
constexpr int INFB_CTX = 8;

struct InfBatch {
	int ctx = -1;                                             // entry of the device list
	hipStream_t st = nullptr;
	Buf h_in{ nullptr, 0, true }, h_out{ nullptr, 0, true }, h_meta{ nullptr, 0, true };
	uint8_t *din = nullptr, *dout = nullptr, *dmeta = nullptr;   // device views of the pinned buffers
	uint32_t state = 0;                                       // futex word: 0 free, 1 collecting, 2 closed (running), 3 done
	int n = 0;
	size_t in_used = 0, out_used = 0;
	uint32_t flags = 0;
	int ready = 0, taken = 0, sleepers = 0, rc = 0;
	// the pinned table, device-visible: in_off u64[S] | out_off u64[S] | in_len, out_cap, out_len, status u32[S]
	uint64_t *in_off() { return (uint64_t *)h_meta.p; }
	uint64_t *out_off() { return in_off() + INFB_SLOTS; }
	uint32_t *in_len() { return (uint32_t *)(out_off() + INFB_SLOTS); }
	uint32_t *out_cap() { return in_len() + INFB_SLOTS; }
	uint32_t *out_len() { return out_cap() + INFB_SLOTS; }
	int32_t *status() { return (int32_t *)(out_len() + INFB_SLOTS); }
};
InfBatch g_infb[INFB_CTX];
std::mutex g_inf_mu;
std::condition_variable g_inf_free;
int g_inf_open = -1;                                          // the collecting batch
int g_inf_active = 0, g_inf_running = 0;                      // callers inside inflate_one / in batches that are closed
int g_inf_inflight = 0, g_inf_max_inflight = 2;               // closed batches not yet done / HIPDEFLATE_INFLATE_INFLIGHT
long g_inf_window_ns = 400000, g_inf_linger_ns = 60000;       // HIPDEFLATE_INFLATE_WINDOW_US / _LINGER_US
int g_inf_failed = 0;
// "Everybody who is inside has joined" is not enough to close a batch: the reference creates its -@N threads one after the
// other (applet/7bgzf.c:330-345), tens of microseconds apart, and a batch that closes as soon as everybody INSIDE has joined
// goes out with the first few of them -- `7bgzf -d -@16` ran 180 launches of 4..8 streams for 1030 blocks, two side by side,
// where 65 launches of 16 do (profiles/r05_dec_trace.txt).  g_inf_peak remembers how many callers have been inside at once:
// a batch is complete when it holds max(inside now, that peak) less those on the device; a batch the LINGER time closes
// short of it takes the peak down a quarter at a time (callers have gone).  A lone caller never waits.
int g_inf_peak = 0;
// A batch is ~1.5 ms on the device whatever it holds (one wavefront per stream, the chip is empty), and the device runs two
// launches side by side but hardly a third (measured on the latency contexts, DESIGN "Hook": 152 us alone, 173 us each for
// two, 266+ for three): so at most two batches are out, and while they are the collecting batch stays open and grows.
// Sixteen callers settle into two groups of eight that alternate; without the cap they drifted apart into batches of one
// to four that queued behind one another (3.0 ms per call where the kernel takes 1.55).

static void infb_sleep(uint32_t *w, uint32_t seen) { syscall(SYS_futex, w, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); }
static void infb_wake_all(uint32_t *w) { syscall(SYS_futex, w, FUTEX_WAKE_PRIVATE, 0x7fffffff, nullptr, nullptr, 0); }
static inline int64_t infb_now()
{
	return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// (g_inf_mu held) a batch context's memory and stream, once
static int infb_create(InfBatch &b, int k)
{
	if (b.st)
		return 0;
	static const bool env_once = [] {
		if (const char *w = getenv("HIPDEFLATE_INFLATE_WINDOW_US"))
			g_inf_window_ns = atol(w) * 1000;
		if (const char *w = getenv("HIPDEFLATE_INFLATE_LINGER_US"))
			g_inf_linger_ns = atol(w) * 1000;
		if (const char *w = getenv("HIPDEFLATE_INFLATE_INFLIGHT"))
			if (atoi(w) >= 1)
				g_inf_max_inflight = atoi(w);
		return true;
	}();
	(void)env_once;
	const int ndev = hipdeflate_device_count();
	b.ctx = ndev > 0 ? k % ndev : 0;
	const OnCtx on(b.ctx);
	if (ensure() || bind_device())
		return HD_E_NODEVICE;
	const size_t meta = (size_t)INFB_SLOTS * (8 + 8 + 4 + 4 + 4 + 4);
	if (b.h_in.reserve(INFB_IN_ARENA + 64) || b.h_out.reserve(INFB_OUT_ARENA + 64) || b.h_meta.reserve(meta) ||
	    hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&b.din, b.h_in.p, 0) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&b.dout, b.h_out.p, 0) != hipSuccess ||
	    hipHostGetDevicePointer((void **)&b.dmeta, b.h_meta.p, 0) != hipSuccess) {
		b.h_in.release();
		b.h_out.release();
		b.h_meta.release();
		if (b.st)
			(void)hipStreamDestroy(b.st);
		b.st = nullptr;
		return HD_E_NOMEM;
	}
	return 0;
}

static void infb_drain()                                      // hipdeflate_shutdown(): idle batch contexts are closed
{
	std::lock_guard<std::mutex> lk(g_inf_mu);
	for (InfBatch &b : g_infb) {
		if (!b.st || __atomic_load_n(&b.state, __ATOMIC_ACQUIRE) != 0)
			continue;
		if (g_all[b.ctx].ready)
			(void)hipSetDevice(g_all[b.ctx].device);
		(void)hipStreamSynchronize(b.st);
		(void)hipStreamDestroy(b.st);
		b.st = nullptr;
		b.h_in.release();
		b.h_out.release();
		b.h_meta.release();
	}
	g_inf_failed = 0;
}

static int inflate_alone(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, uint32_t flags)
{
	uint64_t ioff = 0, ooff = 0;
	uint32_t ilen = (uint32_t)sourceLen, olen = 0;
	uint32_t ocap = *destLen > 0xfffffff0u ? 0xfffffff0u : (uint32_t)*destLen;
	int32_t st = 0;
	int r = batch_inflate_host(source, &ioff, &ilen, 1, dest, &ooff, &ocap, &olen, nullptr, &st, flags);
	if (r)
		return r;
	if (st)
		return st;
	*destLen = olen;
	return 0;
}

static int inflate_one(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen, uint32_t flags)
{
	if (!dest || !destLen || (!source && sourceLen) || sourceLen >= HD_INFLATE_MAX_IN)
		return HD_E_ARG;
	const size_t ocap = *destLen > 0xfffffff0u ? 0xfffffff0u : *destLen;
	const size_t in_need = up16(sourceLen + 4), out_need = up16(ocap);       // (+4: the kernel reads whole dwords)
	if (in_need > INFB_IN_ARENA / 4 || out_need > INFB_OUT_ARENA / 4)
		return inflate_alone(dest, destLen, source, sourceLen, flags);

	/* ---- join the collecting batch, or open one ------------------------------------------------------------- */
	std::unique_lock<std::mutex> lk(g_inf_mu);
	if (g_inf_failed) {
		const int f = g_inf_failed;
		lk.unlock();
		return f == HD_E_NOMEM ? inflate_alone(dest, destLen, source, sourceLen, flags) : f;
	}
	g_inf_active++;
	InfBatch *b = nullptr;
	for (;;) {
		if (g_inf_open >= 0) {
			b = &g_infb[g_inf_open];
			if (b->flags == flags && b->n < (int)INFB_SLOTS && b->in_used + in_need <= INFB_IN_ARENA &&
			    b->out_used + out_need <= INFB_OUT_ARENA)
				break;
			// it cannot take this stream: closed as it stands (its leader finds it so), a new one is opened
			__atomic_store_n(&b->state, 2u, __ATOMIC_RELEASE);
			g_inf_running += b->n;
			g_inf_inflight++;
			g_inf_open = -1;
		}
		int k;
		for (k = 0; k < INFB_CTX && __atomic_load_n(&g_infb[k].state, __ATOMIC_ACQUIRE) != 0; k++)
			;
		if (k < INFB_CTX) {
			b = &g_infb[k];
			if (const int r = infb_create(*b, k)) {
				g_inf_failed = r;
				g_inf_active--;
				lk.unlock();
				return r == HD_E_NOMEM ? inflate_alone(dest, destLen, source, sourceLen, flags) : r;
			}
			b->n = 0;
			b->in_used = b->out_used = 0;
			b->flags = flags;
			b->rc = 0;
			__atomic_store_n(&b->ready, 0, __ATOMIC_RELAXED);
			__atomic_store_n(&b->taken, 0, __ATOMIC_RELAXED);
			__atomic_store_n(&b->state, 1u, __ATOMIC_RELEASE);
			g_inf_open = k;
			break;
		}
		g_inf_free.wait(lk);                              // every batch context is busy: wait for one to drain
	}
	const int idx = b->n;
	const bool leader = idx == 0;
	const size_t my_in = b->in_used, my_out = b->out_used;
	b->in_off()[idx] = my_in;
	b->out_off()[idx] = my_out;
	b->in_len()[idx] = (uint32_t)sourceLen;
	b->out_cap()[idx] = (uint32_t)ocap;
	b->in_used += in_need;
	b->out_used += out_need;
	__atomic_store_n(&b->n, idx + 1, __ATOMIC_RELAXED);
	// everybody who could join has: the callers inside that are not in a closed batch are all here (or the batch is full)
	int peak = __atomic_load_n(&g_inf_peak, __ATOMIC_RELAXED);
	if (g_inf_active > peak)
		__atomic_store_n(&g_inf_peak, peak = g_inf_active, __ATOMIC_RELAXED);
	if (b->n >= (int)INFB_SLOTS || (b->n >= peak - g_inf_running && g_inf_inflight < g_inf_max_inflight)) {
		__atomic_store_n(&b->state, 2u, __ATOMIC_RELEASE);
		g_inf_running += b->n;
		g_inf_inflight++;
		g_inf_open = -1;
	}
	lk.unlock();

	if (sourceLen)
		memcpy((uint8_t *)b->h_in.p + my_in, source, sourceLen);      /* own stream, no lock held */
	__atomic_add_fetch(&b->ready, 1, __ATOMIC_RELEASE);

	if (leader) {
		// others join while the window is open; whoever completes the batch (above) closes it, or the leader does when
		// nobody has joined for the linger time although callers are missing, or when the window is over
		{
			const int64_t t0 = infb_now(), deadline = t0 + g_inf_window_ns, hard = t0 + 20000000;
			int64_t t_last = t0;
			int seen = 1, spins = 0;
			while (__atomic_load_n(&b->state, __ATOMIC_ACQUIRE) == 1) {
				const int64_t t = infb_now();
				const int now_n = __atomic_load_n(&b->n, __ATOMIC_RELAXED);
				if (now_n != seen) {
					seen = now_n;
					t_last = t;
				}
				const bool slot = __atomic_load_n(&g_inf_inflight, __ATOMIC_RELAXED) < g_inf_max_inflight;
				// complete (every caller that is inside and not on the device has joined), or nobody came for the linger
				// time, or the window is over -- and a launch slot is free (20 ms at most, should a batch hang)
				const int running = __atomic_load_n(&g_inf_running, __ATOMIC_RELAXED), peak = __atomic_load_n(&g_inf_peak, __ATOMIC_RELAXED);
				const bool complete = now_n >= std::max(__atomic_load_n(&g_inf_active, __ATOMIC_RELAXED), peak) - running;
				const bool lingered = t >= deadline || t - t_last >= g_inf_linger_ns;
				if (((complete || lingered) && slot) || t >= hard) {
					if (!complete && lingered && slot)       // callers have gone: the peak comes down, a quarter at a time
						__atomic_store_n(&g_inf_peak, std::max(now_n + running, peak - std::max(1, peak / 4)), __ATOMIC_RELAXED);
					break;
				}
				if (!slot && (++spins & 15) == 0)
					sched_yield();                       // two batches are out for a millisecond yet: leave the CPU to their callers
				else
					__builtin_ia32_pause();
			}
		}
		lk.lock();
		if (__atomic_load_n(&b->state, __ATOMIC_RELAXED) == 1) {
			__atomic_store_n(&b->state, 2u, __ATOMIC_RELEASE);
			g_inf_running += b->n;
			g_inf_inflight++;
			g_inf_open = -1;
		}
		const int n = b->n;
		lk.unlock();
		while (__atomic_load_n(&b->ready, __ATOMIC_ACQUIRE) < n)             /* the others are still copying in */
			__builtin_ia32_pause();
		int rc = 0;
		{
			const OnCtx on(b->ctx);
			if (!(rc = ensure()) && !(rc = bind_device())) {
				hd::InflateArgs a;
				a.in = b->din;
				a.in_off = (const uint64_t *)b->dmeta;
				a.out_off = a.in_off + INFB_SLOTS;
				a.in_len = (const uint32_t *)(a.out_off + INFB_SLOTS);
				a.out_cap = a.in_len + INFB_SLOTS;
				a.out_len = (uint32_t *)(a.out_cap + INFB_SLOTS);
				a.status = (int32_t *)(a.out_len + INFB_SLOTS);
				a.nblocks = (uint32_t)n;
				a.out = b->dout;
				a.crc = nullptr;
				a.ct = cur().d_ct;
				a.flags = b->flags;
				hipLaunchKernelGGL(hd::k_inflate_lat, dim3((uint32_t)n), dim3(hd::INF_LAT_THREADS), 0, b->st, a);      // (four wavefronts per stream: hd_inflate_lat.hpp)
				if (hipGetLastError() != hipSuccess || hipStreamSynchronize(b->st) != hipSuccess) {
					fprintf(stderr, "hipdeflate: hip_inflate: the latency kernel did not run\n");
					rc = HD_E_NODEVICE;
				}
			}
		}
		b->rc = rc;
		lk.lock();
		g_inf_running -= n;
		g_inf_inflight--;
		lk.unlock();
		(void)__atomic_exchange_n(&b->state, 3u, __ATOMIC_SEQ_CST);   /* (a full fence: the load of `sleepers` must not pass it) */
		if (__atomic_load_n(&b->sleepers, __ATOMIC_SEQ_CST))
			infb_wake_all(&b->state);
	} else {
		// a batch is about a millisecond of device time: look for a moment (a batch of short streams), then sleep on the word
		const int64_t spin_until = infb_now() + 20000;
		uint32_t st;
		int spins = 0;
		while ((st = __atomic_load_n(&b->state, __ATOMIC_ACQUIRE)) != 3) {
			if ((++spins & 63) == 0 && infb_now() > spin_until) {
				__atomic_add_fetch(&b->sleepers, 1, __ATOMIC_SEQ_CST);
				if (__atomic_load_n(&b->state, __ATOMIC_SEQ_CST) == st)
					infb_sleep(&b->state, st);
				__atomic_sub_fetch(&b->sleepers, 1, __ATOMIC_ACQ_REL);
			} else {
				__builtin_ia32_pause();
			}
		}
	}
	const int rc = b->rc, nb = b->n;
	int ret = rc;
	if (!rc) {
		const int32_t st = b->status()[idx];
		const uint32_t olen = b->out_len()[idx];
		if (st) {
			ret = st;
		} else {
			if (olen)
				memcpy(dest, (const uint8_t *)b->h_out.p + my_out, olen);   /* own output, no lock held */
			*destLen = olen;
		}
	}
	const bool last = __atomic_add_fetch(&b->taken, 1, __ATOMIC_ACQ_REL) == nb;
	lk.lock();
	g_inf_active--;
	if (last) {
		__atomic_store_n(&b->state, 0u, __ATOMIC_RELEASE);                /* drained: the context can collect again */
		g_inf_free.notify_all();
	}
	lk.unlock();
	return ret;
}

void hipdeflate_test_beside(int keep, uint32_t sub_cap)
{
	g_test_beside_keep = keep < 0 ? 0u : keep > 3 ? 3u : (uint32_t)keep;
	hd::wg_sub_test() = sub_cap;
}

#ifdef HD_EMIT_STATS
int hipdeflate_test_emit_stats(uint64_t *out16)   // [0,8) phases of the emit kernel, [8,16) of build_code for the litlen alphabet
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(hd::g_emit_stats), 128));
	return 0;
}
int hipdeflate_test_emit_stats32(uint64_t *out32)   // ... and [16,32): k_emit_wg's block builder
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(hd::g_emit_stats), 256));
	return 0;
}
#endif

#ifdef HD_CLOCK_STAMPS
// diagnostic build: sums of cycles from wave start to the marks of the level-1 / parse kernel (read and cleared)
int hipdeflate_test_clock_marks(uint64_t *out16)
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(hd::g_clk_mark), 128));
	static const uint64_t zero[16] = { 0 };
	HD_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(hd::g_clk_mark), zero, 128));
	return 0;
}
// diagnostic build: per kernel { shader cycles, 100 MHz ticks, waves, - } summed over the waves since the last call
int hipdeflate_test_clock(uint64_t *out16)
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(hd::g_clk), 128));
	static const uint64_t zero[16] = { 0 };
	HD_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(hd::g_clk), zero, 128));
	return 0;
}
#endif

#ifdef HD_INFLATE_STATS
int hipdeflate_test_inflate_stats(uint64_t *out8)
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(hd::g_inf_stats), 64));
	// slots 6, 7 are reused for nothing: the cycle counters follow behind when the caller left room for 16 words
	return 0;
}
int hipdeflate_test_inflate_stats2(uint64_t *out8)
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(hd::g_inf_stats2), 64));
	return 0;
}
int hipdeflate_test_inflate_cycles(uint64_t *out8)
{
	HD_CHECK(hipDeviceSynchronize());
	HD_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(hd::g_inf_cycles), 64));
	return 0;
}
#endif

int hip_inflate(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen)
{
	return inflate_one(dest, destLen, source, sourceLen, 0);
}

int hip_inflate_flush(unsigned char *dest, size_t *destLen, const unsigned char *source, size_t sourceLen)
{
	return inflate_one(dest, destLen, source, sourceLen, hd::INF_FLUSHED);
}

} // extern "C"
