/* wg_live_lanes.c -- research tool (CPU, not product, not oracle): VERDICT r4 item 4 asks to spare k_parse_wg the 16-byte
 * compare of DEAD candidates (first dword first; 8-bit tags; a two-stage compare).  A wavefront executes a way's compare
 * when ANY of its 64 lanes holds a live candidate for that way, so what such a scheme saves is the share of (step, way)
 * pairs in which NO lane does.  This replays the level-6 table of include/hipdeflate_params.h "WORKGROUP LEVELS" (8192
 * buckets x 4 ways, six-byte key HD_HASH_SLOT6, a step's lanes read their buckets as the steps before left them, the highest
 * lane of a step that shares a bucket stores) over a file cut into blocks and counts, per (step, way):
 *   in range   the entry names a position 1..32768 bytes back
 *   live4      ... whose first four bytes equal the lane's
 *   live8      ... whose first eight do (what a second compare stage would have to take)
 * and prints the share of pairs with at least one such lane, and the mean number of such lanes.
 *   gcc -O2 -I include -o /tmp/wg_live_lanes tools/wg_live_lanes.c && /tmp/wg_live_lanes FILE [BLOCK=1048576] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hipdeflate_params.h"

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: wg_live_lanes FILE [BLOCK]\n"); return 2; }
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 1; }
	fseek(f, 0, SEEK_END); size_t total = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
	uint8_t *data = malloc(total + 64);
	if (fread(data, 1, total, f) != total) return 1;
	memset(data + total, 0, 64);
	size_t block = argc > 2 ? (size_t)atol(argv[2]) : 1048576;
	const unsigned ways = 4, nb = HD_WG_BUCKETS(6);
	uint32_t *table = malloc(sizeof(uint32_t) * nb * ways);     /* position + 1, 0 = never written */
	unsigned long long pairs = 0, any_range = 0, any4 = 0, any8 = 0, lanes4 = 0, lanes8 = 0, lanes_range = 0;
	for (size_t b0 = 0; b0 < total; b0 += block) {
		const uint8_t *in = data + b0;
		size_t n = total - b0 < block ? total - b0 : block;
		memset(table, 0, sizeof(uint32_t) * nb * ways);
		for (size_t S = 0; S + 64 <= n; S += 64) {
			uint32_t h[64], old[64][4];
			for (unsigned l = 0; l < 64; l++) {
				size_t p = S + l;
				uint32_t v = 0, vh = 0;
				memcpy(&v, in + p, 4); memcpy(&vh, in + p + 4, 4);
				h[l] = p + HD_LAZY_KEY_BYTES <= n ? HD_HASH_SLOT6(v, vh, nb) : nb;
				for (unsigned k = 0; k < ways; k++)
					old[l][k] = h[l] < nb ? table[h[l] * ways + k] : 0;
			}
			for (unsigned k = 0; k < ways; k++) {
				unsigned r = 0, a = 0, e = 0;
				for (unsigned l = 0; l < 64; l++) {
					size_t p = S + l;
					uint32_t c1 = old[l][k];
					if (!c1 || h[l] >= nb) continue;
					size_t c = c1 - 1;
					if (p - c > HD_WG_WINDOW) continue;
					r++;
					if (!memcmp(in + c, in + p, 4)) { a++; if (p + 8 <= n && !memcmp(in + c, in + p, 8)) e++; }
				}
				pairs++;
				any_range += r != 0; any4 += a != 0; any8 += e != 0;
				lanes_range += r; lanes4 += a; lanes8 += e;
			}
			/* the step's stores: of the lanes that share a bucket the highest stores { itself, the ways - 1 newest before the step } */
			for (unsigned l = 0; l < 64; l++) {
				if (h[l] >= nb) continue;
				int highest = 1;
				for (unsigned m = l + 1; m < 64; m++)
					if (h[m] == h[l]) { highest = 0; break; }
				if (!highest) continue;
				uint32_t *bk = table + h[l] * ways;
				for (unsigned k = ways - 1; k > 0; k--)
					bk[k] = old[l][k - 1];
				bk[0] = (uint32_t)(S + l + 1);
			}
		}
	}
	printf("%s: %zu bytes in blocks of %zu, %u buckets x %u ways, %llu (step, way) pairs\n", argv[1], total, block, nb, ways, pairs);
	printf("pairs with at least one lane whose candidate is in range %.1f %% (%.1f lanes of 64 on average), equal over 4 bytes %.1f %% (%.1f lanes), over 8 bytes %.1f %% (%.1f lanes)\n",
	       100.0 * any_range / pairs, (double)lanes_range / pairs, 100.0 * any4 / pairs, (double)lanes4 / pairs, 100.0 * any8 / pairs, (double)lanes8 / pairs);
	return 0;
}
