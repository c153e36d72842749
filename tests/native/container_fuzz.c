/*
 * container_fuzz.c -- TEST ONLY: mutational fuzz of one container host's READER under AddressSanitizer + UBSan.
 *
 * Built by tests/test_sanitize_hosts.py as
 *     gcc -fsanitize=address,undefined -fno-sanitize-recover=undefined -Dmain=host_main hd_<x>_host.c
 *         stub_containers.c container_fuzz.c
 * i.e. the host's own main() is linked in as host_main and called in a forked child per mutant (no exec: ~1 ms each):
 * the reader parses an offset table / member lengths / an index member / PNG chunk lengths that come from an untrusted
 * file.  The contract: whatever the file holds, the reader exits -- with 0 or with an error code -- and never earns a
 * sanitizer report, a signal or a hang.  Roles fuzzed: applet/7bgzf.c:295-365, 7dictzip.c:318-323, 7razf.c,
 * 7gzinga.c:76-214, 7ciso.c:87, 7daxcr.c:130, 7png.c:296-331 (the reference's readers of the same formats).
 *
 *   container_fuzz <seed file> <mutants> <rng seed> <stdin|file> <host args...>      ("@" in the args = the mutant's path)
 * prints "<n> mutants, <bad> bad"; a bad mutant is kept as <seed file>.bad<k>; exit code 1 if any.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

int host_main(int argc, char **argv);

static uint64_t rng_s;
static uint32_t rnd(void)
{
	rng_s ^= rng_s << 13;
	rng_s ^= rng_s >> 7;
	rng_s ^= rng_s << 17;
	return (uint32_t)(rng_s >> 16);
}

static size_t mutate(const uint8_t *seed, size_t n, uint8_t *out, size_t cap)
{
	memcpy(out, seed, n);
	size_t len = n;
	const unsigned rounds = 1 + rnd() % 3;
	for (unsigned r = 0; r < rounds && len; r++) {
		switch (rnd() % 8) {
		case 0:                                                   /* flip a few bytes */
			for (unsigned k = 1 + rnd() % 8; k; k--)
				out[rnd() % len] ^= (uint8_t)(1u << (rnd() % 8));
			break;
		case 1:                                                   /* random bytes */
			for (unsigned k = 1 + rnd() % 4; k; k--)
				out[rnd() % len] = (uint8_t)rnd();
			break;
		case 2: {                                                 /* an extreme value in a 32-bit field */
			static const uint32_t ext[] = { 0, 1, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffff0u, 0x10000u, 0xffffu, 0x00ffffffu };
			if (len >= 4) {
				const size_t o = rnd() % (len - 3);
				uint32_t v = ext[rnd() % 9];
				if (rnd() & 1)
					v = __builtin_bswap32(v);
				memcpy(out + o, &v, 4);
			}
			break;
		}
		case 3: {                                                 /* a field off by a little */
			if (len >= 4) {
				const size_t o = rnd() % (len - 3);
				uint32_t v;
				memcpy(&v, out + o, 4);
				v += (uint32_t)(rnd() % 33) - 16;
				memcpy(out + o, &v, 4);
			}
			break;
		}
		case 4:                                                   /* truncate */
			len = rnd() % (len + 1);
			break;
		case 5: {                                                 /* the head of the file is where the tables live */
			const size_t head = len < 128 ? len : 128;
			out[rnd() % head] = (uint8_t)rnd();
			break;
		}
		case 6: {                                                 /* duplicate a run */
			const size_t o = rnd() % len, k = 1 + rnd() % 64;
			if (len + k <= cap && o + k <= len) {
				memmove(out + o + k, out + o, len - o);
				len += k;
			}
			break;
		}
		default: {                                                /* the tail: trailers, index members */
			const size_t tail = len < 64 ? len : 64;
			out[len - 1 - rnd() % tail] = (uint8_t)rnd();
			break;
		}
		}
	}
	return len;
}

int main(int argc, char **argv)
{
	if (argc < 6) {
		fprintf(stderr, "usage: %s <seed file> <mutants> <rng seed> <stdin|file> <host args...>\n", argv[0]);
		return 2;
	}
	const char *seed_path = argv[1];
	const int n_mut = atoi(argv[2]);
	rng_s = 0x9E3779B97F4A7C15ull ^ (uint64_t)strtoull(argv[3], NULL, 0);
	const int via_stdin = !strcmp(argv[4], "stdin");
	FILE *f = fopen(seed_path, "rb");
	if (!f) {
		perror(seed_path);
		return 2;
	}
	fseek(f, 0, SEEK_END);
	const size_t n = (size_t)ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *seed = malloc(n + 1), *mut = malloc(n + 4096);
	if (fread(seed, 1, n, f) != n)
		return 2;
	fclose(f);
	char mpath[512];
	snprintf(mpath, sizeof(mpath), "%s.mutant", seed_path);
	int bad = 0;
	for (int it = 0; it < n_mut; it++) {
		const size_t len = it == 0 ? n : mutate(seed, n, mut, n + 4096);      /* mutant 0 = the seed itself: must pass */
		const uint8_t *data = it == 0 ? seed : mut;
		int fd = open(mpath, O_CREAT | O_TRUNC | O_WRONLY, 0600);
		if (fd < 0 || write(fd, data, len) != (ssize_t)len)
			return 2;
		close(fd);
		fflush(NULL);
		const pid_t pid = fork();
		if (pid == 0) {
			alarm(20);
			const int dn = open("/dev/null", O_WRONLY);
			dup2(dn, 1);
			if (!getenv("HD_FUZZ_VERBOSE"))
				dup2(dn, 2);
			if (via_stdin) {
				const int in = open(mpath, O_RDONLY);
				dup2(in, 0);
			}
			char *av[16];
			int ac = 0;
			av[ac++] = (char *)"host";
			for (int k = 5; k < argc && ac < 15; k++)
				av[ac++] = strcmp(argv[k], "@") ? argv[k] : mpath;
			av[ac] = NULL;
			_exit(host_main(ac, av) & 0x7f);
		}
		int st = 0;
		while (waitpid(pid, &st, 0) < 0 && errno == EINTR)
			;
		/* 99 = the sanitizers' exit code (ASAN_OPTIONS / UBSAN_OPTIONS exitcode=99, set by the test) */
		const int failed = WIFSIGNALED(st) || (WIFEXITED(st) && WEXITSTATUS(st) == 99) || (it == 0 && (!WIFEXITED(st) || WEXITSTATUS(st)));
		if (failed) {
			char keep[600];
			snprintf(keep, sizeof(keep), "%s.bad%d", seed_path, bad);
			rename(mpath, keep);
			fprintf(stderr, "mutant %d: %s %d -> %s\n", it, WIFSIGNALED(st) ? "signal" : "exit", WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st), keep);
			bad++;
		}
	}
	unlink(mpath);
	printf("%d mutants, %d bad\n", n_mut, bad);
	return bad ? 1 : 0;
}
