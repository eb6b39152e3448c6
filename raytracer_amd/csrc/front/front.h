/* Internal interfaces of the host front end (SDL/OBJ/PNG readers, blob
 * packing).  Nothing here is exported; the C ABI is in include/rtamd.h. */
#ifndef RTAMD_FRONT_H
#define RTAMD_FRONT_H

#include <stddef.h>
#include <stdint.h>
#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Thread-local error text behind rtamd_last_error(). */
void rtamd_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

/* A loaded Wavefront OBJ mesh (mesh.h:18-36). */
typedef struct FrontMesh {
	int num_vertices, num_normals, num_texcoords, num_triangles;
	double *vertex;      /* 3 per vertex, float-valued */
	double *normal;      /* 3 per normal, NULL when the file has no vn */
	int32_t *triangle;   /* 6 per triangle: v0 v1 v2 n0 n1 n2 (0-based, -1 = absent) */
} FrontMesh;

/* Returns 0 or a negative rtamd_status; fills *mesh (free with front_mesh_free). */
int front_obj_load(const char *path, FrontMesh *mesh);
void front_mesh_free(FrontMesh *mesh);

/* 8-bit RGB or RGBA, non-interlaced PNG -> malloc'ed bytes, top row first.
 * channels receives 3 or 4 (png.bpp in texture.c:31). */
int front_png_load(const char *path, int *width, int *height, int *channels,
		uint8_t **pixels);

/* zlib-wrapped DEFLATE stream -> exactly out_len bytes. */
int front_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

/* ---- jump-ahead in the rand() stream (glibc_rand.c) ----
 * The generator's raw values w[i] obey w[i+31] = w[i+28] + w[i] (mod 2^32), so for
 * c(x) = x^n mod (x^31 - x^28 - 1), with coefficients mod 2^32,
 *     w[n + k] = sum_j c[j] * w[j + k]      for every k >= 0,
 * and draw d of the stream is w[GLIBC_FIRST_DRAW + d] >> 1. */
#define GLIBC_POLY 31
#define GLIBC_FIRST_DRAW 341
/* c = x^n mod p */
void front_glibc_poly_pow(uint64_t n, uint32_t c[GLIBC_POLY]);
/* out = a * b mod p (out may alias neither) */
void front_glibc_poly_mul(const uint32_t a[GLIBC_POLY], const uint32_t b[GLIBC_POLY], uint32_t out[GLIBC_POLY]);
/* w[0 .. count) for a seed */
void front_glibc_window(uint32_t seed, uint32_t *w, int count);

#ifdef __cplusplus
}
#endif
#endif
