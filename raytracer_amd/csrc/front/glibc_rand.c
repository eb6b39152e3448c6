/* The rand() stream the reference draws from (drand, ray.c:14-17; seeded by
 * srand(0x20071208), raytracer.c:74).  glibc's default generator is the
 * TYPE_3 additive feedback generator: r[i] = r[i-31] + r[i-3] (mod 2^32),
 * output r[i] >> 1, primed from the seed by a Lehmer LCG (16807, modulus
 * 2^31-1, computed in Schrage form) and with the first 310 outputs discarded.
 * This is a restatement of that published algorithm so the stream does not
 * depend on which C library the host runs. */
#include <stdlib.h>
#include <string.h>

#include "front.h"

typedef struct GlibcRand {
	uint32_t r[34];
	int idx;            /* position of the next output in the ring */
} GlibcRand;

static void glibc_seed(GlibcRand *g, uint32_t seed)
{
	int32_t r[34];
	if (seed == 0)
		seed = 1;
	r[0] = (int32_t)seed;
	for (int i = 1; i < 31; i++) {
		/* 16807 * r[i-1] mod (2^31 - 1) without overflow */
		int32_t hi = r[i - 1] / 127773;
		int32_t lo = r[i - 1] % 127773;
		int32_t word = 16807 * lo - 2836 * hi;
		if (word < 0)
			word += 2147483647;
		r[i] = word;
	}
	for (int i = 31; i < 34; i++)
		r[i] = r[i - 31];
	for (int i = 0; i < 34; i++)
		g->r[i] = (uint32_t)r[i];
	g->idx = 0;
	/* ring of the last 34 values; element (34+k) is produced from (k+3) and (k+31) */
	for (int i = 0; i < 310; i++) {
		uint32_t v = g->r[(g->idx + 3) % 34] + g->r[(g->idx + 31) % 34];
		g->r[g->idx] = v;
		g->idx = (g->idx + 1) % 34;
	}
}

static int32_t glibc_next(GlibcRand *g)
{
	uint32_t v = g->r[(g->idx + 3) % 34] + g->r[(g->idx + 31) % 34];
	g->r[g->idx] = v;
	g->idx = (g->idx + 1) % 34;
	return (int32_t)(v >> 1);
}

int rtamd_glibc_rand_fill(uint32_t seed, uint64_t first, uint64_t n, int32_t *out)
{
	if (!out && n) {
		rtamd_set_error("rtamd_glibc_rand_fill: NULL output");
		return RTAMD_ERR_ARG;
	}
	GlibcRand g;
	glibc_seed(&g, seed);
	for (uint64_t i = 0; i < first; i++)
		glibc_next(&g);
	for (uint64_t i = 0; i < n; i++)
		out[i] = glibc_next(&g);
	return 0;
}
