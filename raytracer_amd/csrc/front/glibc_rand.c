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

/* the state srand() leaves before the discarded steps */
static void glibc_prime(GlibcRand *g, uint32_t seed)
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
}

static void glibc_seed(GlibcRand *g, uint32_t seed)
{
	glibc_prime(g, seed);
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

/* ---- jump-ahead -------------------------------------------------------------
 * Raw values: u[0..33] is the primed state, u[i] = u[i-31] + u[i-3] for i >= 34
 * (the ring above), draw d = u[344 + d] >> 1.  w[i] = u[i + 3] obeys the recurrence
 * from i = 0 on, hence GLIBC_FIRST_DRAW = 341. */

void front_glibc_window(uint32_t seed, uint32_t *w, int count)
{
	GlibcRand g;
	glibc_prime(&g, seed);
	/* w[i] = u[i + 3]: the first 31 come from the primed state, the rest from the recurrence */
	for (int i = 0; i < count; i++)
		w[i] = i < GLIBC_POLY ? g.r[i + 3] : w[i - GLIBC_POLY] + w[i - 3];
}

/* Reduces a product of degree <= 60 in place: x^m = x^(m-3) + x^(m-31) for m >= 31. */
static void poly_reduce(uint32_t e[2 * GLIBC_POLY - 1])
{
	for (int m = 2 * GLIBC_POLY - 2; m >= GLIBC_POLY; m--) {
		e[m - 3] += e[m];
		e[m - GLIBC_POLY] += e[m];
		e[m] = 0;
	}
}

void front_glibc_poly_mul(const uint32_t a[GLIBC_POLY], const uint32_t b[GLIBC_POLY], uint32_t out[GLIBC_POLY])
{
	uint32_t e[2 * GLIBC_POLY - 1];
	memset(e, 0, sizeof e);
	for (int i = 0; i < GLIBC_POLY; i++)
		for (int j = 0; j < GLIBC_POLY; j++)
			e[i + j] += a[i] * b[j];
	poly_reduce(e);
	memcpy(out, e, sizeof(uint32_t) * GLIBC_POLY);
}

void front_glibc_poly_pow(uint64_t n, uint32_t c[GLIBC_POLY])
{
	uint32_t result[GLIBC_POLY] = { 1 }, base[GLIBC_POLY] = { 0, 1 }, tmp[GLIBC_POLY];
	while (n) {
		if (n & 1) {
			front_glibc_poly_mul(result, base, tmp);
			memcpy(result, tmp, sizeof tmp);
		}
		front_glibc_poly_mul(base, base, tmp);
		memcpy(base, tmp, sizeof tmp);
		n >>= 1;
	}
	memcpy(c, result, sizeof result);
}

int rtamd_glibc_rand_fill(uint32_t seed, uint64_t first, uint64_t n, int32_t *out)
{
	if (!out && n) {
		rtamd_set_error("rtamd_glibc_rand_fill: NULL output");
		return RTAMD_ERR_ARG;
	}
	if (first < 4096) {
		GlibcRand g;
		glibc_seed(&g, seed);
		for (uint64_t i = 0; i < first; i++)
			glibc_next(&g);
		for (uint64_t i = 0; i < n; i++)
			out[i] = glibc_next(&g);
		return 0;
	}
	/* far into the stream: jump to the 31 raw values before draw `first`, then run
	 * the recurrence from there */
	uint32_t c[GLIBC_POLY], w[2 * GLIBC_POLY], ring[GLIBC_POLY];
	front_glibc_poly_pow(GLIBC_FIRST_DRAW + first - GLIBC_POLY, c);
	front_glibc_window(seed, w, 2 * GLIBC_POLY);
	for (int k = 0; k < GLIBC_POLY; k++) {
		uint32_t v = 0;
		for (int j = 0; j < GLIBC_POLY; j++)
			v += c[j] * w[j + k];
		ring[k] = v;                     /* raw value of position first - 31 + k */
	}
	int pos = 0;
	for (uint64_t i = 0; i < n; i++) {
		/* next = value 31 back + value 3 back */
		uint32_t v = ring[pos] + ring[(pos + 28) % GLIBC_POLY];
		ring[pos] = v;
		pos = (pos + 1) % GLIBC_POLY;
		out[i] = (int32_t)(v >> 1);
	}
	return 0;
}
