/*
 * bgzf_hook.c -- the LD_PRELOAD hook: BGZF_METHOD=hip<level> LD_PRELOAD=./libhipdeflate.so samtools ...
 *
 * Same exported symbol, signature and return values as the reference's
 * bgzf_compress (bgzf_compress.c:39-198): htslib's own bgzf_compress is
 * shadowed through the PLT, so samtools/bcftools must link libhts.so
 * (readme.md:9-14).  Differences, all forced by the device:
 *
 *  - The only coder here is "hip" (BGZF_METHOD=hip, hip1 ... hip9; the trailing
 *    digits are the level, parsed as bgzf_compress.c:60-70 does; no digits = level 1,
 *    hip's default as each method has one at :102-112).  BGZF_METHOD unset or empty
 *    is the reference's default, its zlib at level 6 (:54,:102), i.e. hip6: whoever
 *    preloads this library in place of the reference's gets the bytes-per-block class
 *    he had (until late in round 3 the answer was hip1, 0.45 of the input where zlib-6 makes 0.26;
 *    hip6 makes 0.277).  A BGZF_METHOD that
 *    names one of the reference's CPU coders, or an unknown name (which the reference
 *    silently runs as zlib, :54), keeps WRITING: the hip coder runs at the level the
 *    reference would have used for that name -- its digits, else the method's default
 *    of :102-112 (zlib / libdeflate / zlibng / cryptopp / unknown 6, 7zip 2, the others
 *    1), cut to 9 -- and one line on stderr says so.  (Rounds 1-2 returned -1, "coder
 *    missing", for such names: a user with BGZF_METHOD=libdeflate6 left in the
 *    environment got failing writes where the reference compresses.)
 *  - htslib calls the hook once per 0xff00-byte block from each of its worker
 *    threads and waits for the member.  Calls that arrive together share one
 *    latency-mode batch (hipdeflate_lat_*, HD_FRAME_LATENCY: 16 wavefronts per block
 *    at level 1, members framed by the kernel: header, BSIZE, CRC32, ISIZE, i.e.
 *    bgzf_compress.c:191-197).  Every caller copies its own block into the batch's
 *    pinned memory and its own member out of it, outside any lock; the first caller
 *    of a batch (its leader) waits until every caller that is inside the hook and not
 *    in a batch on the device has joined, nobody has joined for HIPDEFLATE_LINGER_US
 *    (8) or HIPDEFLATE_BATCH_US microseconds (default 60) have passed, launches, and
 *    publishes the result; the others spin on the batch's state (HIPDEFLATE_SPIN_US,
 *    default 400, then they sleep on that word; with more callers than cores they sleep at once).  HOOK_CTX batches can be in flight at once (one
 *    collecting, the others on the device).  A lone caller does not wait at all.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>
#include <limits.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include "hipdeflate.h"
#include "hipdeflate_params.h"

#define HOOK_MAX_BATCH 256
#define HOOK_CTX 8
#define HOOK_BLOCK 0xff00u           /* what a latency-mode BGZF slot takes (16 x 4080); htslib's BGZF_BLOCK_SIZE */

struct hook_batch {
	hipdeflate_lat *lat;
	int sleepers;                /* members of THIS batch asleep on `state` (futex) */
	/* state: 0 free, 1 collecting, 2 closed (copies in flight / on the device), 3 done.  Changed under g_mu
	 * (0 -> 1 -> 2, 3 -> 0) or by the batch's leader (2 -> 3); read with acquire loads by spinning members */
	int state;
	int n;                       /* blocks reserved (under g_mu while collecting, fixed afterwards) */
	int ready, taken;            /* blocks copied in / members copied out (atomic counters) */
	int rc;
	uint32_t len[HOOK_MAX_BATCH];
};

/* One ENGINE per (level, frame): the hook's (BGZF members at BGZF_METHOD's level) and, since round 4, one per level and raw
 * frame for the per-block codecs -- hip_deflate / hip_deflate_flush called from the reference's -@N threads
 * (applet/7bgzf.c:211) share launches exactly as htslib's workers do in the hook (16 callers: 3.8 -> 8.6 GB/s at level 1,
 * and 64 callers no longer collapse to 1.2). */
struct hook_eng {
	pthread_mutex_t mu;
	pthread_cond_t cv_free;      /* a context became free */
	struct hook_batch batch[HOOK_CTX];
	int open;                    /* the batch that is collecting, -1 = none */
	int active;                  /* callers inside the engine right now (atomic) */
	int running;                 /* blocks of the batches that are closed and not yet done (under mu) */
	int inflight;                /* under mu; read without by a leader in its window */
	int64_t t_returned;          /* when the last batch came back from the device (atomic; 0 = none yet) */
	int failed;
	int level, frame;            /* level: the hook's (BGZF_METHOD; a codec engine's is its index); frame: HD_FRAME_BGZF, HD_FRAME_RAW or HD_FRAME_RAW_FLUSH */
	int loud;                    /* the hook prints codec errors as the reference does (bgzf_compress.c:163-169) */
};
#define HOOK_ENG_INIT(fr, ld) { .mu = PTHREAD_MUTEX_INITIALIZER, .cv_free = PTHREAD_COND_INITIALIZER, .open = -1, .level = 1, .frame = (fr), .loud = (ld) }
static struct hook_eng g_hook = HOOK_ENG_INIT(HD_FRAME_BGZF, 1);
static struct hook_eng g_codec[10][2] = { [0 ... 9] = { HOOK_ENG_INIT(HD_FRAME_RAW, 0), HOOK_ENG_INIT(HD_FRAME_RAW_FLUSH, 0) } };
static pthread_once_t g_env_once = PTHREAD_ONCE_INIT, g_knobs_once = PTHREAD_ONCE_INIT;
static long g_window_us = 60; /* a leader never waits longer than this for the batch to fill */
static long g_linger_us = 8;  /* ... nor longer than this after the last caller joined */
static long g_spin_us = 400;  /* a member spins this long for its batch before it sleeps */
/* batches on the device at once.  Two run side by side at nearly the price of one at levels 1-2 (hipdeflate_lat_run on 8
 * blocks, level 2: 152 us alone, 173 us each for two), a third and fourth do not (266, 297 us each -- tools/hook_bench.c
 * HOOK_PAR): while two are out, the collecting batch stays open and grows.
 * Round 5: that is the limit for FULL batches (more callers than a batch holds).  A batch that is merely COMPLETE -- every
 * caller that is not on the device has joined -- waits while another is out (g_merge_inflight) and then for that batch's
 * callers to come back (g_rejoin_us): sixteen callers had settled into two groups of eight that took turns, each paying for
 * the other's launch (level 6: two batches of 8 side by side 227-240 us each, one of 16 alone 186: a call 280 -> 215 us) */
static int g_max_inflight = 2;
static int g_merge_inflight = 1;
static int g_merge_callers = 16;  /* ... while at most this many callers are inside the engine; beyond, batches fill by themselves and
                                   * two side by side win (64 callers, level 6: 403 us a call against 520 merged; 4 / 8 callers:
                                   * 233 -> 191 / 224 -> 209 us; 16: 257 -> 250; tools/r05_hook_merge.sh, profiles/r05_hook_merge.txt) */
static long g_rejoin_us = 30; /* after a batch has come back, the collecting one waits this long for the first of its callers */
static int g_batch_target = HOOK_MAX_BATCH;
static int g_ncpu = 1;
/* HIPDEFLATE_HOOK_STATS=1: where the time of a call goes, printed at exit (ns sums; tools/hook_bench.c reads it) */
static int g_stats;
static int64_t st_calls, st_batches, st_blocks, st_copy_in, st_window, st_ready, st_run, st_member_wait, st_copy_out, st_ctx_wait;

__attribute__((destructor)) static void hook_stats_print(void)
{
	if (!g_stats || !st_calls)
		return;
	fprintf(stderr, "hipdeflate hook (%d usable CPUs): %lld calls in %lld batches (%.1f blocks each); us per call: wait for a context %.1f, copy in %.1f, "
		"copy out %.1f, member waits for its batch %.1f; us per batch: leader's window %.1f, others' copies %.1f, device %.1f\n",
		g_ncpu, (long long)st_calls, (long long)st_batches, st_batches ? (double)st_blocks / st_batches : 0.0,
		st_ctx_wait / 1e3 / st_calls, st_copy_in / 1e3 / st_calls, st_copy_out / 1e3 / st_calls,
		st_member_wait / 1e3 / (st_calls - st_batches ? st_calls - st_batches : 1), st_window / 1e3 / (st_batches ? st_batches : 1),
		st_ready / 1e3 / (st_batches ? st_batches : 1), st_run / 1e3 / (st_batches ? st_batches : 1));
}
#define ST_ADD(var, ns) do { if (g_stats) __atomic_add_fetch(&(var), (ns), __ATOMIC_RELAXED); } while (0)

static inline int64_t now_ns(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

/* Members that have waited long enough sleep ON THE BATCH'S STATE WORD (futex): the leader's one FUTEX_WAKE releases
 * all of them at once.  (The condition variable of rounds 1-2 handed its waiters over one by one through its mutex --
 * microseconds each: with 64 callers on 16 cores, where members must sleep to leave the CPUs to the leaders, a 27-block
 * batch took 376 us instead of 115 and the CPU reference won that case, 6.24 GB/s against 4.27.) */
static inline void state_sleep(int *state, int seen)
{
	syscall(SYS_futex, state, FUTEX_WAIT_PRIVATE, seen, NULL, NULL, 0);
}
static inline void state_wake_all(int *state)
{
	syscall(SYS_futex, state, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0);
}

static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#endif
}

/* CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota (a container on a 128-core host
 * with 16 CPUs' worth of quota reports 128 through sysconf: members would spin where they must sleep) */
static int usable_cpus(void)
{
	int n = 0;
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0)
		n = CPU_COUNT(&set);
	if (n <= 0) {
		const long nc = sysconf(_SC_NPROCESSORS_ONLN);
		n = nc > 0 ? (int)nc : 1;
	}
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");              /* cgroup v2: "<quota> <period>" or "max <period>" */
	if (f) {
		long long q = 0, per = 0;
		if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) {
			const int c = (int)((q + per - 1) / per);
			if (c >= 1 && c < n)
				n = c;
		}
		fclose(f);
	} else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {       /* cgroup v1 */
		long long q = -1, per = 100000;
		if (fscanf(f, "%lld", &q) != 1)
			q = -1;
		fclose(f);
		FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
		if (g) {
			if (fscanf(g, "%lld", &per) != 1)
				per = 100000;
			fclose(g);
		}
		if (q > 0 && per > 0) {
			const int c = (int)((q + per - 1) / per);
			if (c >= 1 && c < n)
				n = c;
		}
	}
	const char *o = getenv("HIPDEFLATE_CPUS");
	if (o && *o && atoi(o) > 0)
		n = atoi(o);
	return n;
}

static void parse_env(void)
{
	/* bgzf_compress.c:53-113: name = prefix, level = trailing decimal digits */
	const char *s = getenv("BGZF_METHOD");
	int g_level = 6;                        /* unset / empty: the reference's default is zlib at level 6 (bgzf_compress.c:54,:102) */
	if (s && *s) {
		size_t l = strlen(s), i = l;
		int level = -1, digit = 1;
		while (i > 0 && s[i - 1] >= '0' && s[i - 1] <= '9') {
			if (level < 0)
				level = 0;
			level += digit * (s[i - 1] - '0');
			digit *= 10;
			i--;
		}
		if (i == 3 && !strncasecmp(s, "hip", 3)) {
			g_level = level >= 0 ? level : 1;
		} else {
			/* a name of the reference's table, or an unknown one (its zlib): its level, our coder */
			static const struct { const char *name; int deflt; } ref[] = {
				{ "zlib", 6 }, { "7zip", 2 }, { "7-zip", 2 }, { "zopfli", 1 }, { "miniz", 1 }, { "slz", 1 }, { "libslz", 1 },
				{ "libdeflate", 6 }, { "zlibng", 6 }, { "igzip", 1 }, { "cryptopp", 6 } };
			int deflt = 6;
			for (size_t k = 0; k < sizeof(ref) / sizeof(ref[0]); k++)
				if (strlen(ref[k].name) == i && !strncasecmp(s, ref[k].name, i))
					deflt = ref[k].deflt;
			g_level = level >= 0 ? level : deflt;
			if (g_level > 9)
				g_level = 9;
			fprintf(stderr, "hipdeflate: BGZF_METHOD=%s: this library holds the hip coder only; coding with hip%d\n", s, g_level);
		}
	}
	g_hook.level = g_level;
}

/* the batcher's knobs: read once, by whichever engine runs first (the hook's method above only when the HOOK is first called:
 * a process may have used the codecs long before it sets BGZF_METHOD) */
static void parse_knobs(void)
{
	const char *w = getenv("HIPDEFLATE_BATCH_US");
	if (w && *w)
		g_window_us = atol(w);
	g_ncpu = usable_cpus();
	const char *lg = getenv("HIPDEFLATE_LINGER_US");
	if (lg && *lg)
		g_linger_us = atol(lg);
	const char *hs = getenv("HIPDEFLATE_HOOK_STATS");
	g_stats = hs && *hs && *hs != '0';
	const char *fl = getenv("HIPDEFLATE_INFLIGHT");
	if (fl && atoi(fl) >= 1)
		g_max_inflight = atoi(fl);
	const char *mg = getenv("HIPDEFLATE_MERGE_INFLIGHT");
	if (mg && atoi(mg) >= 1)
		g_merge_inflight = atoi(mg);
	if (g_merge_inflight > g_max_inflight)
		g_merge_inflight = g_max_inflight;
	const char *mc = getenv("HIPDEFLATE_MERGE_CALLERS");
	if (mc && *mc)
		g_merge_callers = atoi(mc);
	const char *rj = getenv("HIPDEFLATE_REJOIN_US");
	if (rj && *rj)
		g_rejoin_us = atol(rj);
	const char *sp = getenv("HIPDEFLATE_SPIN_US");
	if (sp && *sp)
		g_spin_us = atol(sp);
	const char *t = getenv("HIPDEFLATE_BATCH_BLOCKS");
	if (t && *t) {
		g_batch_target = atoi(t);
		if (g_batch_target < 1)
			g_batch_target = 1;
		if (g_batch_target > HOOK_MAX_BATCH)
			g_batch_target = HOOK_MAX_BATCH;
	}
}

/* a block the latency slots do not take (longer than 0xff00 bytes: not from htslib): one ordinary call */
static int code_alone(void *dst, size_t *dlen, const void *src, size_t slen)
{
	uint64_t off = 0;
	uint32_t len = (uint32_t)slen, olen = 0;
	int32_t st = 0;
	unsigned char *tmp = (unsigned char *)malloc(65536);
	if (!tmp)
		return -1;
	int rc = hipdeflate_batch_deflate((const uint8_t *)src, &off, &len, 1, g_hook.level, HD_FRAME_BGZF, tmp, 65536, 65536, &olen,
					  NULL, &st);
	int ret = rc ? -1 : (st || olen > *dlen) ? 1 : 0;
	if (ret == 1)
		fprintf(stderr, "hip_deflate %d\n", st ? st : 1);
	if (!ret) {
		memcpy(dst, tmp, olen);
		*dlen = olen;
	}
	free(tmp);
	return ret;
}

static int eng_compress(struct hook_eng *e, int level, void *_dst, size_t *_dlen, const void *src, size_t slen);

int bgzf_compress(void *_dst, size_t *_dlen, const void *src, size_t slen, int level_unused)
{
	(void)level_unused;
	if (!slen) {
		/* bgzf_compress.c:40-51 */
		if (*_dlen < 28)
			return -1;
		*_dlen = 28;
		memcpy(_dst,
		       "\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff"
		       "\x06\0BC\x02\x00"
		       "\x1b\x00"
		       "\x03\x00"
		       "\x00\x00\x00\x00\x00\x00\x00\x00",
		       28);
		return 0;
	}
	pthread_once(&g_env_once, parse_env);
	pthread_once(&g_knobs_once, parse_knobs);
	if (*_dlen < 26)                            /* bgzf_compress.c:116 */
		return -1;
	if (slen > 0x10000)                         /* a BGZF member cannot hold it */
		return 1;
	if (slen > HOOK_BLOCK || __atomic_load_n(&g_hook.failed, __ATOMIC_RELAXED))
		return g_hook.failed ? -1 : code_alone(_dst, _dlen, src, slen);
	__atomic_add_fetch(&g_hook.active, 1, __ATOMIC_RELAXED);
	const int ret = eng_compress(&g_hook, g_hook.level, _dst, _dlen, src, slen);
	__atomic_sub_fetch(&g_hook.active, 1, __ATOMIC_RELAXED);
	return ret;
}

/* The per-block codecs' way in (hd_api.hip deflate_one): one block of 1 .. 0xff00 bytes whose room covers the latency form's
 * worst case and the stored form -- then the bytes do not depend on the room, and callers with different rooms can share a
 * launch.  0 ok, 1 codec error, -1 the engine cannot run (the caller takes its own context), -2 not for a batch. */
__attribute__((visibility("hidden"))) int hd_codec_batch(unsigned char *dest, size_t *destLen, const unsigned char *src, size_t slen,
							 int level, int flush)
{
	if (level < 0 || level > 9 || !slen || slen > HOOK_BLOCK)
		return -2;
	pthread_once(&g_knobs_once, parse_knobs);
	struct hook_eng *e = &g_codec[level][flush ? 1 : 0];
	if (__atomic_load_n(&e->failed, __ATOMIC_RELAXED))
		return -1;
	__atomic_add_fetch(&e->active, 1, __ATOMIC_RELAXED);
	const int ret = eng_compress(e, level, dest, destLen, src, slen);
	__atomic_sub_fetch(&e->active, 1, __ATOMIC_RELAXED);
	return ret;
}

static int eng_compress(struct hook_eng *e, int level, void *_dst, size_t *_dlen, const void *src, size_t slen)
{
	pthread_mutex_lock(&e->mu);
	if (e->failed) {
		pthread_mutex_unlock(&e->mu);
		return -1;
	}
	/* ---- join the collecting batch, or open one ---------------------------------------------- */
	struct hook_batch *b;
	const int64_t t_enter = g_stats ? now_ns() : 0;
	for (;;) {
		if (e->open >= 0) {
			b = &e->batch[e->open];
			break;
		}
		int k;
		for (k = 0; k < HOOK_CTX && __atomic_load_n(&e->batch[k].state, __ATOMIC_ACQUIRE) != 0; k++)
			;
		if (k < HOOK_CTX) {
			b = &e->batch[k];
			if (!b->lat) {
				/* batch context k lives on entry k of the device list (HIPDEFLATE_DEVICES), round robin */
				const int ndev = hipdeflate_device_count();
				b->lat = hipdeflate_lat_open_on(ndev > 0 ? k % ndev : 0, level, e->frame | HD_FRAME_LATENCY, HOOK_MAX_BATCH,
								HOOK_BLOCK);
				if (!b->lat) {
					e->failed = 1;
					pthread_mutex_unlock(&e->mu);
					return -1;                          /* coder missing */
				}
			}
			__atomic_store_n(&b->n, 0, __ATOMIC_RELAXED);
			__atomic_store_n(&b->ready, 0, __ATOMIC_RELAXED);
			__atomic_store_n(&b->taken, 0, __ATOMIC_RELAXED);
			__atomic_store_n(&b->state, 1, __ATOMIC_RELEASE);
			e->open = k;
			break;
		}
		pthread_cond_wait(&e->cv_free, &e->mu);   /* every context is busy: wait for one to drain */
	}
	const int idx = __atomic_fetch_add(&b->n, 1, __ATOMIC_RELAXED);     /* (written under g_mu; the leader's window loop reads it without) */
	const int leader = idx == 0;
	b->len[idx] = (uint32_t)slen;
	/* Everybody who could join has: the callers inside the hook that are not in a batch on the device are all here
	 * (callers released together by the previous batch come back within microseconds of each other, and count
	 * as inside while they copy their members out).  Or the batch is full. */
	const int want = __atomic_load_n(&e->active, __ATOMIC_RELAXED) - e->running;
	if (b->n >= HOOK_MAX_BATCH || (b->n >= g_batch_target && e->inflight < g_max_inflight) || (b->n >= want && e->inflight < (want + e->running <= g_merge_callers ? g_merge_inflight : g_max_inflight))) {
		__atomic_store_n(&b->state, 2, __ATOMIC_RELEASE);
		e->running += b->n;
		__atomic_add_fetch(&e->inflight, 1, __ATOMIC_RELAXED);
		e->open = -1;
	}
	pthread_mutex_unlock(&e->mu);

	const int64_t t_joined = g_stats ? now_ns() : 0;
	memcpy(hipdeflate_lat_input(b->lat, (uint32_t)idx), src, slen);      /* own block, no lock held */
	__atomic_add_fetch(&b->ready, 1, __ATOMIC_RELEASE);
	const int64_t t_copied = g_stats ? now_ns() : 0;
	ST_ADD(st_calls, 1);
	ST_ADD(st_ctx_wait, t_joined - t_enter);
	ST_ADD(st_copy_in, t_copied - t_joined);

	if (leader) {
		/* Others join while the window is open; whoever completes the batch (above) closes it.  The leader closes
		 * it himself when nobody has joined for g_linger_us although callers are missing (they are busy elsewhere),
		 * or when the window is over. */
		if (g_window_us > 0) {
			const int64_t t0 = now_ns(), deadline = t0 + g_window_us * 1000, hard = t0 + 2000000;
			int64_t t_last = t0;
			int seen = 1;
			while (__atomic_load_n(&b->state, __ATOMIC_ACQUIRE) == 1) {
				const int64_t t = now_ns();
				const int cur = __atomic_load_n(&b->n, __ATOMIC_RELAXED);
				if (cur != seen) {
					seen = cur;
					t_last = t;
				}
				/* (while g_merge_inflight batches are out the window stays open -- 2 ms at most, should one hang --, and
				 * when one has come back its callers get g_rejoin_us to show up, then the linger counts from join to join) */
				const int full = __atomic_load_n(&e->inflight, __ATOMIC_RELAXED) >= (cur >= g_batch_target || __atomic_load_n(&e->active, __ATOMIC_RELAXED) > g_merge_callers ? g_max_inflight : g_merge_inflight);
				const int64_t tr = __atomic_load_n(&e->t_returned, __ATOMIC_RELAXED);
				const int64_t quiet = tr > t_last ? tr + g_rejoin_us * 1000 : t_last + g_linger_us * 1000;
				const int64_t dl = tr > t0 && tr + g_window_us * 1000 > deadline ? tr + g_window_us * 1000 : deadline;
				if ((t >= dl || t >= quiet) && (!full || t >= hard))
					break;
				/* with every device slot taken and more callers than CPUs, the leaders in flight and the HIP runtime's
				 * thread need this CPU more than a spinning window does */
				if (full && __atomic_load_n(&e->active, __ATOMIC_RELAXED) + 1 >= g_ncpu)
					sched_yield();
				else
					cpu_relax();
			}
		}
		pthread_mutex_lock(&e->mu);
		if (__atomic_load_n(&b->state, __ATOMIC_RELAXED) == 1) {          /* window over */
			__atomic_store_n(&b->state, 2, __ATOMIC_RELEASE);
			e->running += b->n;
			__atomic_add_fetch(&e->inflight, 1, __ATOMIC_RELAXED);
			e->open = -1;
		}
		const int n = b->n;
		pthread_mutex_unlock(&e->mu);
		const int64_t t_closed = g_stats ? now_ns() : 0;
		while (__atomic_load_n(&b->ready, __ATOMIC_ACQUIRE) < n)     /* the others are still copying in */
			cpu_relax();
		const int64_t t_ready = g_stats ? now_ns() : 0;
		b->rc = hipdeflate_lat_run(b->lat, b->len, (uint32_t)n);
		__atomic_store_n(&e->t_returned, now_ns(), __ATOMIC_RELAXED);
		pthread_mutex_lock(&e->mu);
		e->running -= n;
		__atomic_sub_fetch(&e->inflight, 1, __ATOMIC_RELAXED);
		pthread_mutex_unlock(&e->mu);
		/* (an exchange, i.e. a full fence: the load of `sleepers` below must not pass this store -- a member that
		 * has counted itself in and still reads state 2 goes to sleep) */
		(void)__atomic_exchange_n(&b->state, 3, __ATOMIC_SEQ_CST);
		if (g_stats) {
			ST_ADD(st_batches, 1);
			ST_ADD(st_blocks, n);
			ST_ADD(st_window, t_closed - t_copied);
			ST_ADD(st_ready, t_ready - t_closed);
			ST_ADD(st_run, now_ns() - t_ready);
		}
		if (__atomic_load_n(&b->sleepers, __ATOMIC_SEQ_CST))
			state_wake_all(&b->state);
	} else {
		/* spin for the batch (all members see it within a cache miss of the leader's store; a condition variable
		 * hands its waiters over one by one, microseconds each), sleep only when it takes long */
		/* as many callers as cores, or more: a spinning (or yielding) member only keeps a leader -- or the HIP runtime's
		 * own thread, which a leader's hipStreamSynchronize waits for -- off its CPU: those members sleep at once (16
		 * callers on 16 CPUs, hip2: 4.51 -> 4.76 GB/s); otherwise a member spins g_spin_us for its batch first (it
		 * sees the leader's store within a cache miss) */
		const int crowded = __atomic_load_n(&e->active, __ATOMIC_RELAXED) + 1 >= g_ncpu;
		const int64_t deadline = crowded ? 0 : now_ns() + g_spin_us * 1000;
		int spins = 0, st;
		while ((st = __atomic_load_n(&b->state, __ATOMIC_ACQUIRE)) != 3) {
			if (crowded || ((++spins & 63) == 0 && now_ns() > deadline)) {
				__atomic_add_fetch(&b->sleepers, 1, __ATOMIC_SEQ_CST);
				/* (the state may be 1 or 2 here; a change to either wakes nobody, so sleep only on what is seen and
				 * look again: FUTEX_WAIT returns at once when the word has moved on) */
				if (__atomic_load_n(&b->state, __ATOMIC_SEQ_CST) == st)
					state_sleep(&b->state, st);
				__atomic_sub_fetch(&b->sleepers, 1, __ATOMIC_ACQ_REL);
			} else {
				cpu_relax();
			}
		}
		if (g_stats)
			ST_ADD(st_member_wait, now_ns() - t_copied);
	}
	const int64_t t_done = g_stats ? now_ns() : 0;
	const int rc = b->rc;
	const int nb = b->n;

	int ret;
	uint32_t olen = 0;
	int32_t st = 0;
	const uint8_t *m = hipdeflate_lat_output(b->lat, (uint32_t)idx, &olen, NULL, &st);
	if (rc || !m) {
		ret = -1;                                   /* coder missing */
	} else if (st || olen > *_dlen) {
		if (e->loud)
			fprintf(stderr, "hip_deflate %d\n", st ? st : 1);
		ret = 1;                                    /* codec error, bgzf_compress.c:163-169 */
	} else {
		memcpy(_dst, m, olen);                      /* own member, no lock held */
		*_dlen = olen;
		ret = 0;
	}
	if (g_stats)
		ST_ADD(st_copy_out, now_ns() - t_done);
	if (__atomic_add_fetch(&b->taken, 1, __ATOMIC_ACQ_REL) == nb) {
		pthread_mutex_lock(&e->mu);
		__atomic_store_n(&b->state, 0, __ATOMIC_RELEASE);   /* drained: the context can collect again */
		pthread_cond_broadcast(&e->cv_free);
		pthread_mutex_unlock(&e->mu);
	}
	return ret;
}
