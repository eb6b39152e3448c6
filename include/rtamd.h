/*
 * rtamd.h -- C ABI of the MI355X render-loop replacement for kaspermeerts/raytracer.
 *
 * The reference has no plugin interface: its drivers call a handful of C
 * functions that read two process globals (scene.h:105-106).  This header is
 * the boundary that takes their place.  Every entry point names the reference
 * interface it replaces.  All pointers are plain host or device pointers; no
 * C++ or torch types appear.
 *
 * Conventions
 *   - every function returns 0 on success and a negative rtamd_status on
 *     failure; rtamd_last_error() returns a message for the calling thread.
 *     (The reference prints to stdout and returns false/NULL, raytracer.c:63-68.)
 *   - "scene blob": one relocatable little-endian byte string that holds the
 *     whole flattened scene (layout below).  It is what sdl_load()'s Sdl/Scene/
 *     Config object graph (scene.h:69-103) becomes on this side of the boundary.
 *   - image layout is the reference's: float RGBA `Colour buffer[width*height]`,
 *     index width*j+i, row j = 0 is the BOTTOM scanline (raytracer.c:72,81).
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum rtamd_status {
	RTAMD_OK = 0,
	RTAMD_ERR_ARG = -1,      /* bad argument / malformed blob */
	RTAMD_ERR_IO = -2,       /* file could not be opened or parsed */
	RTAMD_ERR_SCENE = -3,    /* scene description error (unknown node, missing light ...) */
	RTAMD_ERR_HIP = -4,      /* a HIP runtime call failed */
	RTAMD_ERR_NOMEM = -5,
	RTAMD_ERR_LIMIT = -6,    /* a documented size limit was exceeded */
	RTAMD_ERR_QUEUE = -7     /* a ray queue overflowed (result incomplete) */
};

/* ------------------------------------------------------------------------ *
 * Scene blob layout.  Offsets are bytes from the start of the blob.
 * ------------------------------------------------------------------------ */

#define RTAMD_BLOB_MAGIC 0x324353444d415452ull /* "RTAMDSC2" */
#define RTAMD_MAX_LIGHTS 8                     /* scene.h:10 */

enum rtamd_shape_type {          /* scene.h:47-48, same numbering */
	RTAMD_SHAPE_PLANE = 0,
	RTAMD_SHAPE_DISK = 1,
	RTAMD_SHAPE_SPHERE = 2,
	RTAMD_SHAPE_CYLINDER = 3,
	RTAMD_SHAPE_CONE = 4,
	RTAMD_SHAPE_MESH = 5
};

enum rtamd_light_type { RTAMD_LIGHT_POINT = 0, RTAMD_LIGHT_AREA = 1 }; /* lighting.h:13 */

typedef struct rtamd_light {      /* lighting.h:12-19 */
	int32_t type;
	float intensity;
	double position[3];
	double edge1[3];
	double edge2[3];
	float colour[4];
} rtamd_light;

typedef struct rtamd_material {   /* lighting.h:21-28 */
	float diffuse[4];
	float specular[4];
	int32_t shininess;
	float reflect;
	float glossiness;
	int32_t pad;
} rtamd_material;

/* One placed shape.  Surfaces are stored in the reference's traversal order,
 * i.e. REVERSE document order (scene.c:342-343); the index in this array is
 * the "surface ordinal" reported in hit records. */
typedef struct rtamd_surface {    /* scene.h:46-67 */
	int32_t shape_type;
	int32_t material;             /* index into materials */
	int32_t mesh;                 /* index into meshes, -1 for analytic shapes */
	int32_t capped;
	double edge1[3];              /* plane */
	double edge2[3];
	float radius;                 /* disk, sphere, cylinder, cone */
	float height;                 /* cylinder, cone */
	double model_to_world[16];    /* column-major m[4*col+row], matrix.c:66 */
	double world_to_model[16];
	float bbox[6];                /* xmin ymin zmin xmax ymax zmax, bbox.h:8-11 */
} rtamd_surface;

typedef struct rtamd_mesh {       /* mesh.h:18-36 */
	int32_t num_vertices;
	int32_t num_normals;
	int32_t num_triangles;
	int32_t pad;
	uint64_t off_vertex;          /* double[3*num_vertices] (float-valued, mesh.c:93-95) */
	uint64_t off_normal;          /* double[3*num_normals]; 0 when the OBJ has no vn */
	uint64_t off_triangle;        /* int32[6*num_triangles]: v0 v1 v2 n0 n1 n2, 0-based */
	uint64_t off_name;            /* NUL-terminated name of the <Mesh> shape (Shape.name,
	                               * scene.h:57; printed by scene.c:662); 0 = unnamed */
} rtamd_mesh;

typedef struct rtamd_cubemap {    /* texture.h:14-22 */
	int32_t width[6];             /* face order -x +x -y +y -z +z, texture.h:6-10 */
	int32_t height[6];
	uint64_t off_face[6];         /* float[4*w*h] RGBA, row 0 = bottom (colour.c:49) */
} rtamd_cubemap;

typedef struct rtamd_blob_header {
	uint64_t magic;
	uint64_t total_size;
	/* Config, scene.h:94-103 */
	int32_t width, height;
	int32_t antialiasing, aa_samples;
	int32_t shadow_samples, reflection_samples;
	int32_t max_reflections, depth_of_field;
	/* Camera, scene.h:12-19 */
	double cam_position[3];
	double cam_u[3], cam_v[3], cam_w[3];
	float cam_fov, cam_near_plane;
	/* Scene, scene.h:69-76 */
	float background[4];
	int32_t num_lights;           /* lights in `lights=` order */
	int32_t num_materials;
	int32_t num_surfaces;
	int32_t num_meshes;
	int32_t has_cubemap;
	int32_t pad;
	uint64_t off_lights;          /* rtamd_light[num_lights] */
	uint64_t off_materials;       /* rtamd_material[num_materials] */
	uint64_t off_surfaces;        /* rtamd_surface[num_surfaces] */
	uint64_t off_meshes;          /* rtamd_mesh[num_meshes] */
	uint64_t off_cubemap;         /* rtamd_cubemap, valid if has_cubemap */
} rtamd_blob_header;

/* ------------------------------------------------------------------------ *
 * Front end (host only, no GPU needed).
 * ------------------------------------------------------------------------ */

/* Replaces `Sdl *sdl_load(const char *filename)` (scene.h:108, scene.c:674-704)
 * including the DTD attribute defaults (sdl.dtd), mesh_load (mesh.c:190-231),
 * cubemap_load (texture.c:118-133) and the per-surface build_bbox pass
 * (scene.c:655-658).  Asset paths inside the file are resolved against
 * `asset_dir` (NULL = current directory, which is what the reference does,
 * scene.c:213-214, texture.c:127).  *blob is malloc'ed; free with rtamd_blob_free. */
int rtamd_sdl_load(const char *sdl_path, const char *asset_dir,
		void **blob, size_t *blob_size);
void rtamd_blob_free(void *blob);

/* The reference's loader prints "Loading cubemap took ..." while it runs
 * (scene.c:570-573).  This returns the time the last rtamd_sdl_load of the
 * calling thread spent loading the cubemap, in seconds, or a negative value if
 * that scene has none, so that a driver can print the same line. */
double rtamd_last_cubemap_load_seconds(void);

/* Structural validation of a blob (bounds of every offset and index). */
int rtamd_blob_validate(const void *blob, size_t blob_size);

/* Replaces `bool ppm_write(Colour*,int,int,FILE*)` (ppm.h:8, ppm.c:8-35):
 * P6, top row first, byte = (unsigned char)CLAMP(c*255,0,255). */
int rtamd_ppm_write(const float *colour_rgba, int width, int height, const char *path);
/* The same image as an 8-bit RGB PNG.  The reference writes PPM only; its vendored
 * pnglite carries a writer that is never called (pnglite.c:822-846).  Same
 * quantiser, so the PNG's pixels are the PPM's. */
int rtamd_png_write(const float *colour_rgba, int width, int height, const char *path);
/* Same quantiser into memory: rgb8 has 3*width*height bytes, top row first. */
int rtamd_quantise_rgb8(const float *colour_rgba, int width, int height, uint8_t *rgb8);

/* The glibc TYPE_3 rand() stream behind drand() (ray.c:14-17) after srand(seed)
 * (raytracer.c:74): fills out[0..n) with the values rand() would return for
 * draws first..first+n-1. */
int rtamd_glibc_rand_fill(uint32_t seed, uint64_t first, uint64_t n, int32_t *out);

/* ------------------------------------------------------------------------ *
 * Device side.
 * ------------------------------------------------------------------------ */

typedef struct rtamd_scene rtamd_scene;   /* opaque: device-resident flattened scene */

/* Selects the HIP device for the calling process and creates the library's
 * context.  Must be called before any function below. */
int rtamd_init(int device);
int rtamd_shutdown(void);

/* Flattens a blob into structure-of-arrays device buffers and builds one BVH
 * per mesh.  Replaces the kd-tree pass of import_sdl (scene.c:659-668,
 * mesh.c:417-451); the tree is a different structure with the same nearest-hit
 * semantics. */
int rtamd_scene_create(const void *blob, size_t blob_size, rtamd_scene **scene);
int rtamd_scene_destroy(rtamd_scene *scene);

enum rtamd_rng_mode {
	/* The reference's serial rand() stream indexed in closed form:
	 * draw 2*((j*W+i)*aa^2+k)+{0,1} feeds AA sample k of pixel (i,j).  Exact
	 * only for scenes whose consumption is fixed (no area light, no glossy
	 * material); rejected otherwise.  The draws are generated on the device by
	 * jumping into the stream, one group of tile rows at a time: any frame
	 * size works as long as one row of tiles draws at most 2^28 values. */
	RTAMD_RNG_GLIBC = 0,
	/* Counter-based generator keyed on (seed, pixel, sample, path, purpose,
	 * index): independent of tiling, GPU count and execution order. */
	RTAMD_RNG_KEYED = 1
};

typedef struct rtamd_hit_record {   /* ray.h:14-19 plus the triangle index */
	int32_t surface;                /* surface ordinal, -1 = miss */
	int32_t triangle;               /* index into the mesh's triangle array, -1 if not a mesh */
	float t;                        /* Hit.t (a float-valued double, ray.c:485,574) */
	float pad;
	double position[3];
	double normal[3];
} rtamd_hit_record;

typedef struct rtamd_render_params {
	uint32_t seed;                  /* raytracer.c:74 uses 0x20071208 */
	int32_t rng_mode;               /* rtamd_rng_mode */
	/* Screen-space tiling: the frame is cut into tile_w x tile_h tiles numbered
	 * row-major from the bottom-left; this call renders tiles
	 * tile_first, tile_first+tile_stride, ...  (1 GPU: first=0, stride=1). */
	int32_t tile_w, tile_h;
	int32_t tile_first, tile_stride;
	/* Optional pixel window [x0,x1) x [y0,y1) inside the frame (all zero =
	 * whole frame); pixels outside it are left untouched. */
	int32_t x0, y0, x1, y1;
	int32_t flags;                  /* RTAMD_RENDER_* bits, 0 = none */
	/* Sample subset (with RTAMD_RENDER_PROGRESSIVE): of every pixel's aa_samples^2
	 * samples this call renders sample_count of those numbered sample_first,
	 * sample_first + sample_stride, ... (count 0: all of them).  0 / 0 / 0 = all samples. */
	int32_t sample_first, sample_stride;
	int32_t sample_count;
} rtamd_render_params;

/* rtamd_render without `stats`: bracket every primary-kernel launch with HIP events on
 * the launch stream and keep them with the scene; rtamd_scene_take_kernel_times reads
 * them later.  The call stays asynchronous (no wait for the device per frame). */
#define RTAMD_RENDER_TIME_KERNELS 1
/* Progressive refinement, the role of the reference's rayviewer loop
 * (rayviewer.c:128-150) with antialiasing: the scene keeps the frame's per-pixel sums
 * from call to call; every call adds the samples of its subset (sample_first,
 * sample_stride) and writes the frame scaled by the number of samples summed so far,
 * so each call leaves a complete, noisier picture.  RTAMD_RENDER_RESTART clears the
 * sums first (first pass of a frame).  Once every sample has been rendered the frame
 * is, bit for bit, the one a single call renders.  Where the sample count divides 64
 * (a pixel's samples share a wave and are added in float, in the reference's order,
 * raytracer.c:41-46) the passes must keep that order: sample_stride 1, sample_first =
 * the samples rendered so far, sample_count a divisor of 64.  Other sample counts are
 * summed in 2^-32 fixed point, which commutes: any subsets in any order.
 * Whole frame and all tiles only (tile_stride 1, no window), no hit records. */
#define RTAMD_RENDER_PROGRESSIVE 2
#define RTAMD_RENDER_RESTART 4

typedef struct rtamd_render_stats {
	uint64_t rays_primary;          /* ray_intersect-equivalent queries, by kind */
	uint64_t rays_shadow;           /* shadow queries as shading.c:22-47 makes them; one whose two Phong
	                                 * terms are exactly zero is counted, but answered without a walk
	                                 * (it adds +0 whether occluded or not) */
	uint64_t rays_secondary;
	uint64_t env_lookups;
	uint64_t queue_peak;            /* largest ray-queue population seen */
	uint64_t samples;               /* primary samples generated */
	float ms_total;                 /* device time of the whole call (HIP events) */
	float ms_primary;               /* device time of the primary/shade kernel(s) */
	int32_t launches_primary;
	int32_t launches_bounce;
	uint64_t bvh_nodes;             /* BVH nodes visited / triangle tests made; only */
	uint64_t tri_tests;             /* counted by builds with -DRT_TRAVERSAL_STATS */
} rtamd_render_stats;

/* Replaces the pixel loop raytracer.c:78-84 + pixel_colour (raytracer.c:32-54)
 * and everything below it (ray.c, shading.c, lighting.c, texture.c:72-182).
 *   d_colour : DEVICE pointer, float[4*width*height], reference layout.
 *   d_hits   : DEVICE pointer or NULL; when given, rtamd_hit_record for the
 *              primary ray of every sample of the window, window-local index
 *              ((j-y0)*(x1-x0) + (i-x0))*spp + k (whole frame: (width*j+i)*spp + k).
 *   stream   : hipStream_t (as void*), NULL = default stream.  The call is
 *              asynchronous unless `stats` is non-NULL (then it synchronises
 *              the stream to read counters back).
 * A scene handle supports ONE render in flight: its work buffers are shared by
 * all calls.  A ray-queue overflow (RTAMD_ERR_QUEUE, the frame is incomplete) is
 * returned by the call itself when `stats` is given, otherwise by the next render
 * call on the scene or by rtamd_scene_destroy.
 */
int rtamd_render(rtamd_scene *scene, const rtamd_render_params *params,
		void *d_colour, void *d_hits, rtamd_render_stats *stats, void *stream);

/* rtamd_render for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  Same frame, bit for bit, as rtamd_render
 * with d_hits = NULL and stats = NULL; asynchronous on `stream`.  All work
 * memory (accumulators, ray queues, sort scratch, counters) comes from one
 * static array of 128 MiB that every GPU the library runs on holds until the
 * process ends; the call makes no hipMalloc / hipFree.  Limits, reported as
 * RTAMD_ERR_LIMIT / RTAMD_ERR_ARG: the accumulators of a reflective frame (24
 * bytes per owned pixel) and at least one workgroup's worth of queued rays must
 * fit into the array, and so must the rand() draws of one row of tiles in the
 * RTAMD_RNG_GLIBC mode with antialiasing (8 Mi draws at most).
 * Under stream capture: in the RTAMD_RNG_GLIBC mode with antialiasing the first
 * call for a seed (and the first after a call with another mode) uploads the
 * stream's jump tables from host memory and synchronises the stream, which capture
 * does not allow; that call returns RTAMD_ERR_ARG on a capturing stream.  Make one
 * call with the seed outside capture first.  RTAMD_RNG_KEYED has no such state.
 * Calls on one GPU share the array: calls that overlap in time (other streams,
 * other scenes) race; serialise them.  A queue overflow is reported by the next
 * render call on the scene or by rtamd_scene_destroy. */
int rtamd_render_noalloc(rtamd_scene *scene, const rtamd_render_params *params,
		void *d_colour, void *stream);

/* Replaces `bool ray_intersect(Ray, Hit*)` (ray.h:25, ray.c:643-672) for a
 * batch.  d_rays: DEVICE double[8*n] = origin xyz, direction xyz, near, far
 * (near/far are narrowed to float like Ray.near/far, ray.h:10-11).
 * d_hits: DEVICE rtamd_hit_record[n]. */
int rtamd_trace(rtamd_scene *scene, const void *d_rays, uint64_t n,
		void *d_hits, void *stream);

/* Replaces camera_ray / camera_ray_aa (ray.h:23-24, ray.c:19-65) for a batch:
 * sample k of pixel (i,j) for all pixels of the window/tiles selected by
 * `params`; d_rays receives double[8] per sample at (width*j+i)*spp + k. */
int rtamd_camera_rays(rtamd_scene *scene, const rtamd_render_params *params,
		void *d_rays, void *stream);

/* Multi-GPU framebuffer exchange helpers (device pointers).  pack copies the
 * tiles owned by (tile_first, tile_stride) from a frame-layout buffer into a
 * dense [n_owned_tiles][tile_h][tile_w][4] float buffer; unpack is the inverse
 * for the tiles of the given owner.  rtamd_tiles_owned returns the tile count. */
int rtamd_tiles_owned(int width, int height, const rtamd_render_params *params);
int rtamd_tiles_pack(int width, int height, const rtamd_render_params *params,
		const void *d_frame, void *d_packed, void *stream);
int rtamd_tiles_unpack(int width, int height, const rtamd_render_params *params,
		const void *d_packed, void *d_frame, void *stream);

/* rtamd_tiles_unpack for the owners first_owner .. tile_stride-1 in ONE launch: owner
 * r's dense buffer starts at d_packed_all + r * owner_stride_bytes (the layout a gather
 * of equal-size per-rank buffers leaves on rank 0; params->tile_first is ignored).
 * Has no counterpart in the reference (its frame is written by one loop, raytracer.c:78-84). */
int rtamd_tiles_unpack_all(int width, int height, const rtamd_render_params *params, int first_owner,
		const void *d_packed_all, uint64_t owner_stride_bytes, void *d_frame, void *stream);

/* Host-side query helpers. */
/* Device time of the primary-kernel launches recorded since the last call by renders
 * with RTAMD_RENDER_TIME_KERNELS (waits for the last of them): their sum in ms and
 * their number.  The measurement behind the reference's "Rendering took ..." line
 * (raytracer.c:76,89-90), taken per kernel launch. */
int rtamd_scene_take_kernel_times(rtamd_scene *scene, float *ms_primary, int *launches);
/* Seconds rtamd_scene_create spent on mesh `mesh`'s acceleration structure (the
 * figure behind the reference's "Building kd-tree took ..." line, scene.c:663-666). */
int rtamd_scene_mesh_build_seconds(const rtamd_scene *scene, int mesh, double *seconds);
/* Whether mesh `mesh`'s hierarchy is the one built by the GPU kernels (1) or by the
 * host's builder (0): the host builds meshes below 2^18 triangles, and any mesh whose
 * GPU-built tree came out deeper than the traversal stack.  Takes the place of the
 * reference's single kd-tree builder (mesh.c:417-451); no effect on results. */
int rtamd_scene_mesh_built_on_device(const rtamd_scene *scene, int mesh, int *on_device);
int rtamd_scene_dims(const rtamd_scene *scene, int *width, int *height, int *spp);
const char *rtamd_last_error(void);
const char *rtamd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
