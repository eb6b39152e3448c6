/*
 * oracle_cli.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Command-line front of the CPU restatement: renders a scene blob and writes
 * the same files as _ref/ref_harness (<prefix>.colour.f32, optional
 * <prefix>.hits.bin), and prints counters and wall time as one JSON line.
 * Used by tests and by the cpu_baseline leg of bench.py.
 *
 *   oracle_cli render <blob> <prefix> [--rng glibc|keyed] [--seed S]
 *              [--threads N] [--window x0 y0 x1 y1] [--hits] [--size W H]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oracle.h"

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
	if (argc < 4 || strcmp(argv[1], "render")) {
		fprintf(stderr, "usage: oracle_cli render <blob> <prefix> [--rng glibc|keyed] "
				"[--seed S] [--threads N] [--window x0 y0 x1 y1] [--hits] [--size W H]\n");
		return 2;
	}
	FILE *f = fopen(argv[2], "rb");
	if (!f) {
		fprintf(stderr, "cannot open %s\n", argv[2]);
		return 2;
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	unsigned char *blob = malloc((size_t)n);
	if (fread(blob, 1, (size_t)n, f) != (size_t)n)
		return 2;
	fclose(f);

	unsigned seed = 0x20071208;
	int rng = 0, threads = 1, want_hits = 0;
	int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
	for (int a = 4; a < argc; a++) {
		if (!strcmp(argv[a], "--rng") && a + 1 < argc)
			rng = !strcmp(argv[++a], "keyed");
		else if (!strcmp(argv[a], "--seed") && a + 1 < argc)
			seed = (unsigned)strtoul(argv[++a], NULL, 0);
		else if (!strcmp(argv[a], "--threads") && a + 1 < argc)
			threads = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--hits"))
			want_hits = 1;
		else if (!strcmp(argv[a], "--window") && a + 4 < argc) {
			x0 = atoi(argv[++a]);
			y0 = atoi(argv[++a]);
			x1 = atoi(argv[++a]);
			y1 = atoi(argv[++a]);
		} else if (!strcmp(argv[a], "--size") && a + 2 < argc) {
			rtamd_blob_header *h = (rtamd_blob_header *)blob;
			h->width = atoi(argv[++a]);
			h->height = atoi(argv[++a]);
		} else {
			fprintf(stderr, "unknown option %s\n", argv[a]);
			return 2;
		}
	}
	oracle_scene *s = oracle_scene_create(blob, (size_t)n);
	if (!s) {
		fprintf(stderr, "bad blob\n");
		return 2;
	}
	const rtamd_blob_header *h = (const rtamd_blob_header *)blob;
	int w = h->width, hh = h->height;
	int spp = h->antialiasing ? h->aa_samples * h->aa_samples : 1;
	int wx0 = x0, wy0 = y0, wx1 = x1 ? x1 : w, wy1 = y1 ? y1 : hh;
	size_t nsamples = (size_t)(wx1 - wx0) * (size_t)(wy1 - wy0) * (size_t)spp;
	float *colour = calloc((size_t)w * hh * 4, sizeof(float));
	rtamd_hit_record *hits = want_hits ? calloc(nsamples, sizeof *hits) : NULL;
	oracle_stats st;
	double t0 = now();
	int rc = oracle_render(s, seed, rng, x0, y0, x1, y1, colour, hits, &st, threads);
	double t1 = now();
	if (rc) {
		fprintf(stderr, "oracle_render failed\n");
		return 1;
	}
	char path[2048];
	snprintf(path, sizeof path, "%s.colour.f32", argv[3]);
	f = fopen(path, "wb");
	fwrite(colour, sizeof(float), (size_t)w * hh * 4, f);
	fclose(f);
	if (hits) {
		snprintf(path, sizeof path, "%s.hits.bin", argv[3]);
		f = fopen(path, "wb");
		fwrite(hits, sizeof *hits, nsamples, f);
		fclose(f);
	}
	unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_secondary;
	printf("{\"seconds\": %.6f, \"rays\": %llu, \"rays_primary\": %llu, \"rays_shadow\": %llu, "
			"\"rays_secondary\": %llu, \"env_lookups\": %llu, \"light_samples\": %llu, "
			"\"draws\": %llu, \"bbox_tests\": %llu, \"surface_probes\": %llu, "
			"\"mesh_probes\": %llu, \"mesh_probe_levels\": %llu, "
			"\"threads\": %d, \"mrays_per_s\": %.6f}\n",
			t1 - t0, rays, (unsigned long long)st.rays_primary,
			(unsigned long long)st.rays_shadow, (unsigned long long)st.rays_secondary,
			(unsigned long long)st.env_lookups, (unsigned long long)st.light_samples,
			(unsigned long long)st.draws, (unsigned long long)st.bbox_tests,
			(unsigned long long)st.surface_probes, (unsigned long long)st.mesh_probes,
			(unsigned long long)st.mesh_probe_levels, threads, rays / (t1 - t0) / 1e6);
	oracle_scene_destroy(s);
	return 0;
}
