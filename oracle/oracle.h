/*
 * oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the reference's render path (ray.c, shading.c,
 * lighting.c, texture.c:72-182, raytracer.c:32-84) working on a scene blob
 * (include/rtamd.h).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py compares this
 * restatement bit for bit (float colours, hit records, stream position)
 * against the reference's own sources compiled by oracle/Makefile
 * (_ref/ref_harness) wherever /root/reference exists, and everywhere against
 * the vectors under tests/golden/ that were generated the same way.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

typedef struct oracle_stats {
	uint64_t rays_primary;     /* ray_intersect calls, by caller */
	uint64_t rays_shadow;
	uint64_t rays_secondary;
	uint64_t env_lookups;      /* cubemap_colour calls */
	uint64_t light_samples;    /* unoccluded light samples that were shaded */
	uint64_t draws;            /* drand() calls */
	uint64_t bbox_tests;       /* ray_bbox_test calls */
	uint64_t surface_probes;   /* ray_surface_intersect calls */
	uint64_t mesh_probes;      /* ... of which on a mesh surface */
	uint64_t mesh_probe_levels;/* sum over mesh probes of ceil(log2(triangles / 4)) */
} oracle_stats;

/* Builds the oracle's own acceleration structure (a plain double-precision
 * BVH; the reference's kd-tree shape is not observable in its output). */
oracle_scene *oracle_scene_create(const void *blob, size_t blob_size);
void oracle_scene_destroy(oracle_scene *s);

/* rng_mode 0: the reference's serial rand() stream, consumed in the
 *   reference's order (raytracer.c:74-84); single-threaded by nature.  With a
 *   window, pixels outside it are skipped by discarding 2*aa^2 draws each,
 *   which is exact only for fixed-consumption scenes.
 * rng_mode 1: keyed counter generator (same definition as the HIP kernels);
 *   `threads` > 1 splits rows over threads.
 * colour: float[4*width*height], reference layout (row 0 = bottom), only
 *   pixels inside the window are written.  Alpha is not computed (set to 0).
 * hits: NULL or one record per sample of the window, window-local order
 *   ((j-y0)*(x1-x0)+(i-x0))*spp+k.
 * Window all-zero = whole frame. */
int oracle_render(const oracle_scene *s, uint32_t seed, int rng_mode,
		int x0, int y0, int x1, int y1, float *colour, rtamd_hit_record *hits,
		oracle_stats *stats, int threads);

/* ray_intersect (ray.c:643-672) on a batch: rays = double[8*n]
 * (origin, direction, near, far). */
int oracle_trace(const oracle_scene *s, const double *rays, size_t n,
		rtamd_hit_record *hits);

/* camera_ray / camera_ray_aa (ray.c:19-65) with explicit jitter values:
 * offsets = float[2*n] (offx, offy), pixels = int32[2*n] (i, j);
 * rays out = double[8*n]. */
int oracle_camera_rays(const oracle_scene *s, const int32_t *pixels, const float *offsets,
		double near_plane, size_t n, double *rays);

/* Nearest triangle of mesh `m` by exhaustive search (no BVH), model space;
 * used to check the BVH traversal.  ray = double[8]. Returns triangle or -1. */
int oracle_mesh_bruteforce(const oracle_scene *s, int m, const double *ray, double *t_out);

/* Debugging aid: one pixel in serial-stream mode with every ray_intersect call
 * logged (8 doubles per ray); returns the number of rays. */
size_t oracle_debug_pixel_rays(const oracle_scene *s, uint32_t seed, int x, int y, double *rays,
		size_t cap, float *colour_out);

/* The keyed generator, exposed so tests can pin its definition. */
uint32_t oracle_keyed_key0(uint32_t seed, uint32_t pixel, uint32_t sample);
uint32_t oracle_keyed_child(uint32_t key, uint32_t child);
uint32_t oracle_keyed_draw(uint32_t key, uint32_t purpose, uint32_t index);

#ifdef __cplusplus
}
#endif
#endif
