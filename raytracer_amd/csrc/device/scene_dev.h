/* Device-side scene layout (structure-of-arrays buffers in HBM) shared by the
 * host flattening code and the HIP kernels.  Internal to librtamd. */
#ifndef RTAMD_SCENE_DEV_H
#define RTAMD_SCENE_DEV_H

#include <stdint.h>
#include "rtamd.h"

#define RT_MAX_STACK 32        /* BVH traversal stack entries per lane (LDS) */
#define RT_BLOCK 256           /* threads per workgroup: 4 wave64 */
#define RT_FIX_SCALE 4294967296.0   /* 2^32: fixed-point scale of the accumulators */

/* One placed shape (scene.h:60-67) with everything the intersectors need
 * precomputed in the reference's own arithmetic (same operations, same order,
 * so the bits are the ones the reference computes per ray). */
struct DevSurface {
	float bbox[6];             /* xmin ymin zmin xmax ymax zmax */
	int32_t shape_type;
	int32_t material;
	int32_t mesh;
	int32_t capped;
	int32_t affine;            /* world_to_model row 3 is exactly (0,0,0,1) */
	int32_t linear_identity;   /* affine and the 3x3 part is exactly I (zeros +0) */
	float radius, height;
	double w2m[16];            /* column-major, matrix.c:66 */
	/* plane (ray.c:69-97): n = e1 x e2, axn = e1 x n, bxn = e2 x n, det = e1 . bxn */
	double pn[3], paxn[3], pbxn[3];
	float pdet;
	float pad1;
};

struct DevMaterial {
	float diffuse[3];
	float specular[3];
	int32_t shininess;
	float reflect;
	float glossiness;
	int32_t pad;
};

struct DevLight {
	int32_t type;
	float intensity;
	double position[3];
	double edge1[3];
	double edge2[3];
	float colour[3];
	float pad;
};

/* BVH node: two children with their boxes, 64 bytes, read as four float4.
 *   q0 = lo0.x lo0.y lo0.z hi0.x
 *   q1 = hi0.y hi0.z lo1.x lo1.y
 *   q2 = lo1.z hi1.x hi1.y hi1.z
 *   q3 = bits(child0) bits(child1) 0 0
 * child >= 0: inner node index (relative to the mesh's first node);
 * child <  0: leaf, v = ~child, first triangle = v >> 2, count = (v & 3) + 1. */
struct DevBvhNode {
	float q[16];
};

/* Leaf-ordered triangle: three float4 (v0, v1, v2); v0.w carries the bits of
 * the triangle's index in the mesh's own triangle array. */
struct DevTriangle {
	float v[12];
};

/* Leaf-ordered triangle for the packet traversal: first vertex and the two edge
 * vectors of ray.c:334-335 (edge = v - u evaluated in double on the host, the
 * same single rounding the reference performs per ray), plus the triangle's
 * index in the mesh's own array.  80 bytes, fetched with scalar loads. */
struct DevTriEdges {
	double u[3];
	double e1[3];
	double e2[3];
	int32_t orig;
	int32_t pad[1];
};

struct DevMesh {
	int32_t node_first;        /* into nodes[] */
	int32_t tri_first;         /* into tris[] / normals[] */
	int32_t root;              /* child reference of the root (leaf if < 0) */
	int32_t num_triangles;
};

struct DevCubeFace {
	int32_t width, height;
	int64_t offset;            /* in float4 texels from texels[0] */
};

/* Everything a kernel needs, passed by value (fits the kernarg segment). */
struct DevScene {
	int32_t width, height;
	int32_t antialiasing, aa_samples;
	int32_t shadow_samples, reflection_samples;
	int32_t max_reflections;
	int32_t spp;
	double cam_pos[3], cam_u[3], cam_v[3], cam_w[3];
	/* image-plane extents from cam_ray_internal (ray.c:27-30) for the two
	 * `near` values the drivers pass (raytracer.c:43,49), computed on the host
	 * because they involve tan() */
	double cam_width, cam_left, cam_height, cam_bottom;
	float cam_d;               /* `d = near` narrowed to float, ray.c:32 */
	float background[3];
	int32_t num_lights, num_surfaces, has_cubemap, any_reflective;
	int32_t any_area_light;
	int32_t shadow_rays_per_hit; /* point lights + shadow_samples^2 per area light */
	/* the samples of a pixel one call renders: sub_count of them, numbered sub_first +
	 * sub_stride * i (all spp of them: 0, 1, spp); resolve_samples = samples summed in
	 * the accumulators when the frame is written (spp unless the frame is refined
	 * progressively) */
	int32_t sub_first, sub_stride, sub_count, resolve_samples;
	const DevLight *lights;
	const DevSurface *surfaces;
	const DevMaterial *materials;
	const DevMesh *meshes;
	const DevBvhNode *nodes;
	const DevTriangle *tris;
	const DevTriangle *normals; /* same order as tris: n0, n1, n2 */
	const DevTriEdges *edges;   /* same order as tris */
	const float *texels;        /* float4 RGBA */
	DevCubeFace faces[6];
};

/* A ray waiting in a bounce queue.  depth == QUEUE_SLOT_UNUSED marks a slot that
 * was reserved but never filled (queue overflow): consumers skip it. */
#define QUEUE_SLOT_UNUSED 0xffffffffu
struct QueueItem {
	double o[3];
	double d[3];
	float weight[3];
	uint32_t slot;              /* accumulator slot (pixel) */
	uint32_t key;               /* keyed-RNG path key */
	uint32_t depth;
	uint32_t pad[2];
};

/* The part of a bounce queue one launch takes: entries [first, first + limit) of the
 * queue's population (read on the device, so the host never waits for it). */
struct QueueSlice {
	const QueueItem *items;
	const unsigned *count;      /* the queue's population */
	unsigned capacity;          /* entries the queue holds; a larger population was cut off */
	unsigned first, limit;
};

/* Nearest hit of a queued ray, kept between the trace and the shade kernel of a
 * bounce pass (the position and normal are rebuilt from it, ray.c:574-576). */
struct BounceHit {
	int32_t surface;            /* -1: miss (already shaded) or unused slot */
	int32_t tri;
	int32_t ncode;
	float t;
	float a, b, c;
	uint32_t pad;
};

#define RT_SORT_BITS_MAX 10     /* bits per axis of the hit-point grid (30-bit Morton keys) */
#define RT_SORT_DIGIT 10        /* bits per pass of the radix sort of the hits */
#define RT_SORT_RADIX (1 << RT_SORT_DIGIT)

/* A run of the reference's rand() stream in device memory: draws[i] is draw
 * number first + i. */
struct GlibcTable {
	const int32_t *draws;
	unsigned long long first;
};

#define RT_GLIBC_POLY 31        /* degree of the generator's recurrence */
#define RT_GLIBC_BLOCK 4096     /* draws produced by one workgroup of the fill kernel */
#define RT_GLIBC_FIRST_DRAW 341 /* raw-sequence position of draw 0 */

/* x^n mod (x^31 - x^28 - 1) with coefficients mod 2^32: where a fill starts */
struct GlibcJump {
	uint32_t c[RT_GLIBC_POLY];
};

/* The counter block exists RT_COUNTER_COPIES times, 128 bytes apart.  A wave adds
 * its ray counts to the copy its workgroup number selects, so that the millions of
 * waves of a frame do not queue up on four addresses (atomics on one address are
 * worked off one after the other, a few nanoseconds each: at 4K x 64 spp that
 * alone was most of a frame of planes).  Copy 0 also holds the rare events
 * (overflow, peak); k_collect_stats folds the others into it. */
#define RT_COUNTER_COPIES 256
struct __attribute__((aligned(128))) DevCounters {
	unsigned long long rays_primary, rays_shadow, rays_secondary, env_lookups;
	unsigned long long queue_overflow;
	unsigned long long queue_peak;
	unsigned long long bvh_nodes;      /* only counted in RT_TRAVERSAL_STATS builds */
	unsigned long long tri_tests;
};

#endif
