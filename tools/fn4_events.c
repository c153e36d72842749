/* fn4_events.c -- research tool (CPU, not product, not oracle): what the "one-dword automaton with continuation lanes"
 * (round 4's lead for the level-1 kernel) would cost in SCALAR events.
 *
 * The level-1 kernel resolves the greedy parse of a step's 64 positions with a prefix scan over transition functions on the
 * states 0..7 ("lanes still covered by the current match"): eight bytes per lane, two v_perm_b32 + two DPP moves per stage,
 * six stages = 24 slow vector instructions (hd_device.hpp fn8_scan).  With FOUR states a function is one dword and a stage
 * one v_perm + one DPP: 12.  A match of 5..8 bytes does not fit four states; it is taken apart as the kernel already takes
 * apart its 9..15-byte matches (hd_deflate_static.hpp "K16, continuation lanes"): the parent lane jumps 4 and lane + 4 --
 * its continuation lane -- jumps the rest, whatever candidate that lane has itself.  That is exact when the parse takes the
 * parent.  When the parse arrives at a continuation lane WITHOUT having taken its parent (the parent lies under an earlier
 * token), the lane's own candidate counts and the chain behind it has to be re-threaded by scalar code: an EVENT
 * (~12 scalar instructions + 6 per hop of the re-threading walk in the shipped kernel's K16 form).
 *
 * This tool replays level 1's step parse (the twin's matchfinder, #included: same table, same window, same candidates) and
 * runs both resolutions per step: the exact greedy one and the four-state one with the kernel's event loop, and prints events
 * and re-threading hops per step, and the same for the shipped split at 8 (events there = matches of 16 bytes and more +
 * orphaned continuation lanes) for comparison.
 *
 *   gcc -O2 -I include -o /tmp/fn4_events tools/fn4_events.c && /tmp/fn4_events FILE [BLOCK]
 */
#define main twin_unused_main
#include "../oracle/hd_deflate_twin.c"
#undef main
#include <stdio.h>

typedef struct {
	unsigned long long steps, events, hops, parents, orphan_possible, tokens;
} tally_t;

/* one step: jump[l] = candidate's common length capped at CAP2 = 2 * K (what the lanes know before any scalar work), 1 = no match
 * K = states of the automaton (4 or 8).  Returns the carry into the next step from the EXACT parse. */
static void resolve(const unsigned *full, unsigned lanes, unsigned carry, unsigned K, tally_t *t)
{
	uint64_t par = 0, contm = 0, big = 0, starts = 0;
	unsigned jumpA[64], jumpW[64];
	/* parents: K < len < 2K, not themselves the continuation lane of a parent; big: >= 2K (extended by scalar code when taken) */
	for (unsigned l = 0; l < lanes; l++) {
		unsigned len = full[l];
		jumpW[l] = len < 2 * K ? len : K;                       /* 2K and more: K for now, extended at its event */
		jumpA[l] = len < K ? len : K;
		if (len >= 2 * K)
			big |= 1ull << l;
		else if (len > K && !(l >= K && ((par >> (l - K)) & 1)))
			par |= 1ull << l;
	}
	for (unsigned l = 0; l + K < lanes; l++)
		if ((par >> l) & 1) {
			contm |= 1ull << (l + K);
			jumpA[l + K] = full[l] - K;
		}
	/* the scan's answer: the automaton's chain from `carry` */
	for (unsigned x = carry; x < lanes; x += jumpA[x])
		starts |= 1ull << x;
	uint64_t pend = ~0ull;
	for (;;) {
		uint64_t v = starts & contm & ((starts & par) << K);
		uint64_t ev = starts & (big | contm) & ~v & pend;
		if (!ev)
			break;
		unsigned m = (unsigned)__builtin_ctzll(ev);
		unsigned len = ((big >> m) & 1) ? full[m] : jumpW[m];
		t->events++;
		uint64_t fresh = 0, old = starts & ~contm;
		unsigned xq = m + len;
		while (xq < lanes && !((old >> xq) & 1)) {
			fresh |= 1ull << xq;
			xq += jumpW[xq];
			t->hops++;
		}
		unsigned xe = xq < 64 ? xq : 64;
		uint64_t gone = xe - m - 1 ? (((xe - m - 1 >= 64) ? ~0ull : ((1ull << (xe - m - 1)) - 1)) << (m + 1)) : 0;
		starts = (starts & ~gone) | fresh;
		pend = m == 63 ? 0 : ~1ull << m;
	}
	t->parents += (unsigned)__builtin_popcountll(par);
	t->tokens += (unsigned)__builtin_popcountll(starts & ~(starts & contm & ((starts & par) << K)));
	t->steps++;
}

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: fn4_events FILE [BLOCK=65280]\n");
		return 2;
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 1; }
	fseek(f, 0, SEEK_END);
	size_t total = (size_t)ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *data = malloc(total + 64);
	if (fread(data, 1, total, f) != total) return 1;
	fclose(f);
	size_t block = argc > 2 ? (size_t)atol(argv[2]) : 65280;
	tally_t t4 = { 0 }, t8 = { 0 };
	unsigned long long hist[10] = { 0 }, exact_tokens = 0;
	for (size_t b0 = 0; b0 < total; b0 += block) {
		const uint8_t *in = data + b0;
		size_t n = total - b0 < block ? total - b0 : block;
		mf_t mf = { calloc((size_t)1 << HD_L1_HASH_BITS, 2), HD_L1_HASH_BITS, 1u << HD_L1_WIN_BITS, 0, 0 };
		unsigned carry = 0;
		for (size_t S = 0; S < n; S += HD_WAVE) {
			unsigned lanes = n - S < HD_WAVE ? (unsigned)(n - S) : HD_WAVE;
			uint32_t cand[HD_WAVE];
			unsigned full[HD_WAVE];
			while (mf.filled < n && mf.filled < S + HD_LOOKAHEAD)
				mf.filled += HD_PIECE;
			size_t lo = mf.filled > mf.win ? mf.filled - mf.win : 0;
			for (unsigned l = 0; l < lanes; l++) {
				size_t p = S + l;
				cand[l] = 0;
				if (p + HD_MIN_MATCH > n)
					continue;
				uint32_t e = mf.table[mf_index(&mf, load32(in + p))];
				uint32_t back = (uint32_t)(p + 1 - e) & 0xffffu;
				cand[l] = (e && back) ? (uint32_t)(p + 1 - back) : 0;
			}
			for (unsigned l = 0; l < lanes; l++) {
				size_t p = S + l;
				if (p + HD_MIN_MATCH <= n)
					mf.table[mf_index(&mf, load32(in + p))] = (uint16_t)(p + 1);
			}
			for (unsigned l = 0; l < lanes; l++) {
				size_t p = S + l;
				full[l] = 1;
				if (p + HD_MIN_MATCH > n || cand[l] == 0)
					continue;
				size_t c = cand[l] - 1;
				if (c < lo || load32(in + c) != load32(in + p))
					continue;
				unsigned maxlen = n - p < HD_MAX_MATCH ? (unsigned)(n - p) : HD_MAX_MATCH, len = HD_MIN_MATCH;
				while (len < maxlen && in[p + len] == in[c + len])
					len++;
				full[l] = len;
			}
			if (carry < lanes) {
				resolve(full, lanes, carry, 4, &t4);
				resolve(full, lanes, carry, 8, &t8);
			}
			/* the exact parse, for the carry and the length histogram of the tokens it takes */
			unsigned E = carry;
			for (unsigned l = 0; l < lanes; l++) {
				if (l < E)
					continue;
				E = l + full[l];
				exact_tokens++;
				hist[full[l] < 9 ? full[l] : 9]++;
			}
			carry = E > lanes ? E - lanes : 0;
		}
		free(mf.table);
	}
	printf("%s: %zu bytes, blocks of %zu, level-1 geometry (window %u, %u table entries)\n", argv[1], total, block, 1u << HD_L1_WIN_BITS,
	       HD_TABLE_ENTRIES(HD_L1_WIN_BITS, HD_L1_HASH_BITS));
	printf("tokens of the exact parse by length: literal %.1f %%, 4: %.1f %%, 5..8: %.1f %%, 9+: %.1f %%  (%.2f tokens per step)\n",
	       100.0 * hist[1] / exact_tokens, 100.0 * hist[4] / exact_tokens, 100.0 * (hist[5] + hist[6] + hist[7] + hist[8]) / exact_tokens,
	       100.0 * hist[9] / exact_tokens, (double)exact_tokens / t8.steps);
	printf("(tokens of the exact parse: %llu -- both resolutions below must arrive at the same number)\n", exact_tokens);
	printf("eight states (shipped):  %.3f events per step, %.3f hops per step, %.2f parents per step, tokens %llu\n", (double)t8.events / t8.steps,
	       (double)t8.hops / t8.steps, (double)t8.parents / t8.steps, t8.tokens);
	printf("four states (one dword): %.3f events per step, %.3f hops per step, %.2f parents per step, tokens %llu\n", (double)t4.events / t4.steps,
	       (double)t4.hops / t4.steps, (double)t4.parents / t4.steps, t4.tokens);
	return 0;
}
