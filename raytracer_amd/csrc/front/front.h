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

#ifdef __cplusplus
}
#endif
#endif
