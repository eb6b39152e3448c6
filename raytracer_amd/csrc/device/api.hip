// Device half of the C ABI (include/rtamd.h): context, scene flattening and
// upload, and the host-side orchestration of the render pipeline.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rtamd.h"
#include "scene_dev.h"
#include "kernels.h"
#include "bvh_build.h"
#include "bvh_device.h"
#include "front.h"

extern "C" void rtamd_set_error(const char *fmt, ...);

#define TWO_PI 6.283185307179586476925287   /* cgmath.h:8 */

#define HIP_TRY(expr)                                                              \
	do {                                                                           \
		hipError_t e_ = (expr);                                                    \
		if (e_ != hipSuccess) {                                                    \
			rtamd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
					__FILE__, __LINE__);                                           \
			return RTAMD_ERR_HIP;                                                  \
		}                                                                          \
	} while (0)

namespace {

int g_device = -1;

// Rays per bounce queue (80 B each; three queues and 48 B of sort scratch per entry: 77 GB
// at the limit, a quarter of the GPU's memory).  Large queues are not only fewer
// launches: the more hits a bounce level holds, the closer together lie the 64 hits
// that share a wave after sorting, and the cheaper their shadow rays become (8-glossy at
// 640 x 640 x 64 samples: 233 ms in four chunks, 190 ms in one).
const unsigned QUEUE_CAPACITY_MAX = 1u << 28;
const unsigned QUEUE_CAPACITY_LIMIT = 1u << 28;
const unsigned QUEUE_CAPACITY_MIN = 1u << 16;
const unsigned long long GLIBC_TABLE_DRAWS_DEFAULT = 1ull << 24;   // serial-stream table: 64 MiB
const unsigned long long GLIBC_TABLE_DRAWS_MAX = 1ull << 28;

} // namespace

struct rtamd_scene {
	DevScene dev;
	std::vector<void *> allocations;
	rtamd_blob_header header;
	bool fixed_consumption;      // no area light, no glossy+reflective material in use
	bool glossy;                 // some material in use spawns reflection_samples children
	// image-plane extents for the two `near` values the drivers pass
	// lazily created work buffers
	// serial-stream mode: a bounded table of draws, refilled per group of tile rows by
	// k_glibc_fill from the jump polynomials and the seed's first raw values
	int32_t *d_glibc = nullptr;
	unsigned long long glibc_capacity = 0;       // draws the table holds
	unsigned long long glibc_capacity_max = 0;   // RTAMD_GLIBC_TABLE_DRAWS (test knob) or the default
	uint32_t *d_glibc_powers = nullptr;          // 2048 x 31 words
	uint32_t *d_glibc_window = nullptr;          // RT_GLIBC_BLOCK + 31 words, for window_seed
	uint32_t glibc_window_seed = 0;
	bool glibc_window_valid = false;
	bool glibc_powers_uploaded = false;
	long long *d_accum = nullptr;
	unsigned long long accum_slots = 0;
	float *d_fsum = nullptr;                     // progressive frames: ordered float sums, float4 per pixel
	unsigned long long fsum_slots = 0;
	QueueItem *d_queue[3] = { nullptr, nullptr, nullptr };   // level 1; deeper levels alternate between the other two
	unsigned queue_capacity = 0;
	unsigned *d_qcount = nullptr;
	DevCounters *d_counters = nullptr;
	// coherent bounce pass (see kernels.hip): per-entry scratch sized like the queues
	BounceWork bounce = { nullptr, { nullptr, nullptr }, { nullptr, nullptr }, nullptr, nullptr, nullptr, 0, 0, 0, 1, 0xffffffffu };
	unsigned bounce_capacity = 0;
	// Point lights of intensity zero are left out of the device's light list (see
	// rtamd_scene_create): each is one shadow query per shaded hit, which the kernels do not
	// count; a shaded hit sends `kept_shadow_rays_per_hit` counted ones.
	int unlit_point_lights = 0, kept_shadow_rays_per_hit = 0;
	bool sorted_bounce = false;  // area lights present: bounced hits are shaded in spatial order
	bool bounce_packets = false; // unsorted pass: queued rays walk as wave-wide packets (mirror-only scenes)
	int device = -1;             // the device the scene's buffers live on
	std::vector<double> mesh_build_seconds;
	int progress_samples = 0;    // RTAMD_RENDER_PROGRESSIVE: samples per pixel summed in the accumulators
	// RTAMD_RENDER_TIME_KERNELS: event pairs around primary launches, read by rtamd_scene_take_kernel_times
	std::vector<hipEvent_t> deferred_events;
	std::vector<char> mesh_built_on_device;   // per mesh: its tree is the GPU-built one
	unsigned queue_capacity_max = 0;
	unsigned overflow_test_capacity = 0;
	// queue-overflow count of the last asynchronous render, in host-visible memory
	unsigned long long *h_overflow = nullptr, *d_overflow = nullptr;
};

namespace {

template <typename T>
int upload(rtamd_scene *sc, const std::vector<T> &host, const T **dev)
{
	*dev = nullptr;
	size_t bytes = host.size() * sizeof(T);
	void *p = nullptr;
	HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
	sc->allocations.push_back(p);
	if (bytes)
		HIP_TRY(hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice));
	*dev = static_cast<const T *>(p);
	return 0;
}

void cross3(const double a[3], const double b[3], double c[3])   // vector.c:56-64
{
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}

double dot3(const double a[3], const double b[3])                 // vector.c:31-34
{
	return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

bool float_valued(double v)
{
	return (double)(float)v == v;
}

// Triangles, corner normals and edge records of one mesh in the leaf order of a
// host-built tree.  pos / nrm: 9 floats per triangle in the mesh's own order.
void emit_leaf_ordered(const float *pos, const float *nrm, int nt, const BvhResult &bvh,
		std::vector<DevTriangle> &tris, std::vector<DevTriangle> &normals, std::vector<DevTriEdges> &edges)
{
	for (int i = 0; i < nt; i++) {
		int orig = bvh.order[i];
		DevTriangle tp, tn;
		std::memset(&tp, 0, sizeof tp);
		std::memset(&tn, 0, sizeof tn);
		for (int c = 0; c < 3; c++)
			for (int a = 0; a < 3; a++) {
				tp.v[4 * c + a] = pos[9 * (size_t)orig + 3 * c + a];
				tn.v[4 * c + a] = nrm[9 * (size_t)orig + 3 * c + a];
			}
		std::memcpy(&tp.v[3], &orig, 4);
		tris.push_back(tp);
		normals.push_back(tn);
		DevTriEdges te;
		std::memset(&te, 0, sizeof te);
		for (int a = 0; a < 3; a++) {
			// vec3_sub(v, u), vec3_sub(w, u): ray.c:334-335, vector.c:21-29
			const double u = pos[9 * (size_t)orig + a];
			te.u[a] = u;
			te.e1[a] = (double)pos[9 * (size_t)orig + 3 + a] - u;
			te.e2[a] = (double)pos[9 * (size_t)orig + 6 + a] - u;
		}
		te.orig = orig;
		edges.push_back(te);
	}
}

} // namespace

static int scratch_init(int device);

extern "C" int rtamd_init(int device)
{
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		rtamd_set_error("rtamd_init: no HIP device available (%s)",
				e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
		return RTAMD_ERR_HIP;
	}
	if (device < 0 || device >= count) {
		rtamd_set_error("rtamd_init: device %d out of range (have %d)", device, count);
		return RTAMD_ERR_ARG;
	}
	HIP_TRY(hipSetDevice(device));
	int rc = scratch_init(device);
	if (rc)
		return rc;
	g_device = device;
	return 0;
}

extern "C" int rtamd_shutdown(void)
{
	g_device = -1;
	return 0;
}

extern "C" const char *rtamd_version(void)
{
	return "rtamd 0.1 (gfx950)";
}

static int take_sticky_overflow(rtamd_scene *sc);

static void free_bounce_work(rtamd_scene *sc)
{
	BounceWork &w = sc->bounce;
	void *ptrs[] = { w.hits, w.keys[0], w.keys[1], w.index[0], w.index[1], w.counts, w.totals, w.bounds };
	for (void *p : ptrs)
		if (p)
			(void)hipFree(p);
	w.hits = nullptr;
	w.keys[0] = w.keys[1] = w.index[0] = w.index[1] = nullptr;
	w.counts = w.totals = w.bounds = nullptr;
	sc->bounce_capacity = 0;
}

extern "C" int rtamd_scene_destroy(rtamd_scene *sc)
{
	if (!sc)
		return 0;
	if (sc->device >= 0)
		(void)hipSetDevice(sc->device);
	for (void *p : sc->allocations)
		(void)hipFree(p);
	for (hipEvent_t e : sc->deferred_events)
		(void)hipEventDestroy(e);
	if (sc->d_glibc) (void)hipFree(sc->d_glibc);
	if (sc->d_glibc_powers) (void)hipFree(sc->d_glibc_powers);
	if (sc->d_glibc_window) (void)hipFree(sc->d_glibc_window);
	if (sc->d_accum) (void)hipFree(sc->d_accum);
	if (sc->d_fsum) (void)hipFree(sc->d_fsum);
	for (QueueItem *q : sc->d_queue)
		if (q)
			(void)hipFree(q);
	if (sc->d_qcount) (void)hipFree(sc->d_qcount);
	if (sc->d_counters) (void)hipFree(sc->d_counters);
	free_bounce_work(sc);
	int rc = 0;
	if (sc->h_overflow) {
		// the last asynchronous render may still be running
		(void)hipDeviceSynchronize();
		rc = take_sticky_overflow(sc);
		(void)hipHostFree(sc->h_overflow);
	}
	delete sc;
	return rc;
}

static int scene_create_impl(const uint8_t *base, rtamd_scene *sc)
{
	const rtamd_blob_header &h = sc->header;
	DevScene &d = sc->dev;
	std::memset(&d, 0, sizeof d);
	d.width = h.width;
	d.height = h.height;
	d.antialiasing = h.antialiasing;
	d.aa_samples = h.aa_samples;
	d.shadow_samples = h.shadow_samples;
	d.reflection_samples = h.reflection_samples;
	d.max_reflections = h.max_reflections;
	d.spp = h.antialiasing ? h.aa_samples * h.aa_samples : 1;
	d.sub_first = 0;
	d.sub_stride = 1;
	d.sub_count = d.spp;
	d.resolve_samples = d.spp;
	std::memcpy(d.cam_pos, h.cam_position, sizeof d.cam_pos);
	std::memcpy(d.cam_u, h.cam_u, sizeof d.cam_u);
	std::memcpy(d.cam_v, h.cam_v, sizeof d.cam_v);
	std::memcpy(d.cam_w, h.cam_w, sizeof d.cam_w);
	{
		// ray.c:27-32; `near` is cam->near_plane with AA and the literal 1 without
		// (raytracer.c:43,49)
		double near = h.antialiasing ? (double)h.cam_near_plane : 1.0;
		double width = 2 * near * tan(h.cam_fov * TWO_PI / 360. / 2.);
		d.cam_width = width;
		d.cam_left = -width / 2;
		d.cam_height = width * h.height / (double)h.width;
		d.cam_bottom = -d.cam_height / 2;
		d.cam_d = (float)near;
	}
	d.background[0] = h.background[0];
	d.background[1] = h.background[1];
	d.background[2] = h.background[2];
	d.num_surfaces = h.num_surfaces;
	d.has_cubemap = h.has_cubemap;

	const rtamd_light *bl = reinterpret_cast<const rtamd_light *>(base + h.off_lights);
	const rtamd_material *bm = reinterpret_cast<const rtamd_material *>(base + h.off_materials);
	const rtamd_surface *bs = reinterpret_cast<const rtamd_surface *>(base + h.off_surfaces);
	const rtamd_mesh *bmesh = reinterpret_cast<const rtamd_mesh *>(base + h.off_meshes);

	// A point light of intensity zero (2-transform, 3-aa and 5-primitives have one) gives every
	// sample the terms Kd * (0 * (N.L * Lc)) and Ks * (0 * (pow(H.N, e) * Lc)) (lighting.c:7-34):
	// exactly zero when colours are finite and exponents not negative, whether the shadow ray
	// (shading.c:39-47) is occluded or not.  Such a light is left out of the device's list; its
	// shadow queries are added to the count when it is read (as many as shaded hits, which is
	// the kernels' count over the queries per hit of the lights that stay; the last light
	// always stays).  (Area lights stay: in the serial-stream mode their samples draw from
	// rand() whatever their intensity.)
	bool plain_materials = true;
	for (int i = 0; i < h.num_materials; i++) {
		plain_materials = plain_materials && bm[i].shininess >= 0;
		for (int c = 0; c < 3; c++)
			plain_materials = plain_materials && std::isfinite(bm[i].diffuse[c]) && std::isfinite(bm[i].specular[c]);
	}
	const char *keep_unlit = getenv("RTAMD_KEEP_UNLIT_LIGHTS");      // tests: walk them like any other
	const bool may_drop = plain_materials && !(keep_unlit && atoi(keep_unlit));
	bool drop[RTAMD_MAX_LIGHTS] = { false };
	sc->unlit_point_lights = 0;
	sc->kept_shadow_rays_per_hit = 0;
	for (int i = 0; i < h.num_lights; i++) {
		// an area light's keyed sample sequence is named by its place in the list: only point
		// lights behind the last area light go, so that no place changes
		bool area_behind = false;
		for (int k = i + 1; k < h.num_lights; k++)
			area_behind = area_behind || bl[k].type == RTAMD_LIGHT_AREA;
		drop[i] = may_drop && !area_behind && bl[i].type == RTAMD_LIGHT_POINT && bl[i].intensity == 0.0f
			&& std::isfinite(bl[i].colour[0]) && std::isfinite(bl[i].colour[1]) && std::isfinite(bl[i].colour[2]);
		if (drop[i])
			sc->unlit_point_lights++;
		else
			sc->kept_shadow_rays_per_hit += bl[i].type == RTAMD_LIGHT_AREA ? h.shadow_samples * h.shadow_samples : 1;
	}
	for (int i = 0; i < h.num_lights && sc->kept_shadow_rays_per_hit == 0; i++)
		if (drop[i]) {
			// the count of shaded hits needs one light that is walked
			drop[i] = false;
			sc->unlit_point_lights--;
			sc->kept_shadow_rays_per_hit = 1;
		}
	std::vector<DevLight> lights;
	bool any_area = false;
	for (int i = 0; i < h.num_lights; i++) {
		if (drop[i])
			continue;
		lights.emplace_back();
		DevLight &l = lights.back();
		std::memset(&l, 0, sizeof l);
		l.type = bl[i].type;
		l.intensity = bl[i].intensity;
		std::memcpy(l.position, bl[i].position, sizeof l.position);
		std::memcpy(l.edge1, bl[i].edge1, sizeof l.edge1);
		std::memcpy(l.edge2, bl[i].edge2, sizeof l.edge2);
		l.colour[0] = bl[i].colour[0];
		l.colour[1] = bl[i].colour[1];
		l.colour[2] = bl[i].colour[2];
		any_area |= l.type == RTAMD_LIGHT_AREA;
	}
	d.num_lights = (int)lights.size();
	std::vector<DevMaterial> mats(h.num_materials);
	for (int i = 0; i < h.num_materials; i++) {
		DevMaterial &m = mats[i];
		std::memset(&m, 0, sizeof m);
		for (int c = 0; c < 3; c++) {
			m.diffuse[c] = bm[i].diffuse[c];
			m.specular[c] = bm[i].specular[c];
		}
		m.shininess = bm[i].shininess;
		m.reflect = bm[i].reflect;
		m.glossiness = bm[i].glossiness;
	}

	std::vector<DevSurface> surfs(h.num_surfaces);
	bool any_reflective = false, any_glossy = false;
	for (int i = 0; i < h.num_surfaces; i++) {
		DevSurface &s = surfs[i];
		std::memset(&s, 0, sizeof s);
		std::memcpy(s.bbox, bs[i].bbox, sizeof s.bbox);
		s.shape_type = bs[i].shape_type;
		s.material = bs[i].material;
		s.mesh = bs[i].mesh;
		s.capped = bs[i].capped;
		s.radius = bs[i].radius;
		s.height = bs[i].height;
		std::memcpy(s.w2m, bs[i].world_to_model, sizeof s.w2m);
		s.affine = s.w2m[3] == 0.0 && s.w2m[7] == 0.0 && s.w2m[11] == 0.0 && s.w2m[15] == 1.0;
		{
			bool ident = s.affine != 0;
			for (int c = 0; c < 3 && ident; c++)
				for (int r = 0; r < 3; r++) {
					double e = s.w2m[4 * c + r];
					if (r == c ? e != 1.0 : (e != 0.0 || std::signbit(e)))
						ident = false;
				}
			s.linear_identity = ident;
		}
		if (s.shape_type == RTAMD_SHAPE_PLANE) {
			// ray.c:76-82: n = a x b, axn = a x n, bxn = b x n, det = a . bxn
			cross3(bs[i].edge1, bs[i].edge2, s.pn);
			cross3(bs[i].edge1, s.pn, s.paxn);
			cross3(bs[i].edge2, s.pn, s.pbxn);
			s.pdet = (float)dot3(bs[i].edge1, s.pbxn);
		}
		const rtamd_material &m = bm[s.material];
		if (m.reflect > 0.0f) {
			any_reflective = true;
			if (m.glossiness > 0.0f)
				any_glossy = true;
		}
	}
	{
		// The coherent bounce pass pays when a bounced hit sends many shadow rays
		// (area lights); RTAMD_BOUNCE=sorted|legacy overrides the choice, and
		// RTAMD_SORT_BITS the grid resolution (bits per axis).
		const char *mode = getenv("RTAMD_BOUNCE");
		sc->sorted_bounce = any_area && h.shadow_samples >= 2;
		if (mode && !strcmp(mode, "sorted")) sc->sorted_bounce = true;
		if (mode && !strcmp(mode, "legacy")) sc->sorted_bounce = false;
		const char *bits = getenv("RTAMD_SORT_BITS");
		int b = bits ? atoi(bits) : 9;
		sc->bounce.bits = b < 1 ? 1 : (b > RT_SORT_BITS_MAX ? RT_SORT_BITS_MAX : b);
		// Levels of fewer hits than this (RTAMD_SMALL_LEVEL, 0 = never) are shaded with each
		// hit's light samples across the lanes of a wave: as many hits to a wave as have
		// their n x n samples side by side in 64 lanes.
		const char *small = getenv("RTAMD_SMALL_LEVEL");
		sc->bounce.small_below = small ? (unsigned)strtoul(small, nullptr, 0) : 262144u;
		// the unsorted pass walks bounced rays as packets where every reflection is a mirror
		// and a wave holds the samples of one pixel (measured on 4-environmentmap: 1024^2 x
		// 64 samples 12.5 -> 10.9 ms, but 512^2 x 16 samples 1.10 -> 1.25 ms);
		// RTAMD_BOUNCE_WALK=packet|lane overrides
		const char *walk = getenv("RTAMD_BOUNCE_WALK");
		sc->bounce_packets = !any_glossy && d.spp >= 64;
		if (walk && !strcmp(walk, "packet")) sc->bounce_packets = true;
		if (walk && !strcmp(walk, "lane")) sc->bounce_packets = false;
		// Levels of at least this many hits (RTAMD_PACKETS_FROM; RTAMD_BOUNCE_SHADE=packet: all
		// that are not "small", =lane: none) are shaded with wave-wide shafts and packets.
		const char *from = getenv("RTAMD_PACKETS_FROM");
		sc->bounce.packets_from = from ? (unsigned)strtoul(from, nullptr, 0) : (1u << 18);
		const char *shade = getenv("RTAMD_BOUNCE_SHADE");
		if (shade && !strcmp(shade, "packet")) sc->bounce.packets_from = 0;
		if (shade && !strcmp(shade, "lane")) sc->bounce.packets_from = 0xffffffffu;
		sc->bounce.sort_chunks_max = 2048;
		const int m = h.shadow_samples * h.shadow_samples;
		int per_wave = m >= 64 ? 1 : 64 / m;
		sc->bounce.hits_per_wave = (unsigned)(per_wave > 8 ? 8 : per_wave);
	}
	sc->fixed_consumption = !any_area && !any_glossy;
	sc->glossy = any_glossy;
	d.any_reflective = any_reflective && h.max_reflections >= 1;
	d.any_area_light = any_area && h.shadow_samples >= 2;
	d.shadow_rays_per_hit = 0;
	for (int i = 0; i < h.num_lights; i++)
		d.shadow_rays_per_hit += bl[i].type == RTAMD_LIGHT_AREA ? h.shadow_samples * h.shadow_samples : 1;

	// Meshes: leaf-ordered triangle and normal arrays, one BVH each.
	std::vector<DevMesh> meshes(h.num_meshes);
	std::vector<DevBvhNode> nodes;
	std::vector<DevTriangle> tris, normals;
	std::vector<DevTriEdges> edges;
	// Which meshes get their BVH built on the GPU: RTAMD_BVH=device all of them,
	// RTAMD_BVH=host none; by default those with at least 2^18 triangles, where the
	// host's SAH build (about a microsecond per triangle and level) starts to cost
	// more than the frames it speeds up.
	const char *bvh_mode = getenv("RTAMD_BVH");
	int device_min = 1 << 18;
	if (const char *dm = getenv("RTAMD_BVH_DEVICE_MIN")) {      // test knob: the threshold itself
		const int v = atoi(dm);
		if (v >= 1)
			device_min = v;
	}
	auto device_build = [&](int ntri) {
		if (bvh_mode && !strcmp(bvh_mode, "device")) return true;
		if (bvh_mode && !strcmp(bvh_mode, "host")) return false;
		return ntri >= device_min;
	};
	struct PendingBuild {
		int mesh;
		std::vector<float> pos, nrm;
	};
	std::vector<PendingBuild> pending;
	for (int m = 0; m < h.num_meshes; m++) {
		const rtamd_mesh &bmh = bmesh[m];
		const double *v = reinterpret_cast<const double *>(base + bmh.off_vertex);
		const double *n = reinterpret_cast<const double *>(base + bmh.off_normal);
		const int32_t *t = reinterpret_cast<const int32_t *>(base + bmh.off_triangle);
		const int nt = bmh.num_triangles;
		if (nt <= 0) {
			rtamd_set_error("mesh %d has no triangles", m);
			return RTAMD_ERR_ARG;
		}
		if (nt > (1 << 26)) {
			rtamd_set_error("mesh %d has %d triangles; the limit is %d", m, nt, 1 << 26);
			return RTAMD_ERR_LIMIT;
		}
		for (int i = 0; i < 3 * bmh.num_vertices; i++)
			if (!float_valued(v[i])) {
				rtamd_set_error("mesh %d: vertex coordinate %.17g is not a float value; mesh "
						"coordinates must be float-representable (objreader narrows them, "
						"mesh.c:93-95)", m, v[i]);
				return RTAMD_ERR_ARG;
			}
		for (int i = 0; i < 3 * bmh.num_normals; i++)
			if (!float_valued(n[i])) {
				rtamd_set_error("mesh %d: normal component is not a float value", m);
				return RTAMD_ERR_ARG;
			}
		std::vector<float> pos(9 * (size_t)nt), nrm(9 * (size_t)nt);
		for (int i = 0; i < nt; i++)
			for (int c = 0; c < 3; c++)
				for (int a = 0; a < 3; a++) {
					pos[9 * (size_t)i + 3 * c + a] = (float)v[3 * (size_t)t[6 * i + c] + a];
					nrm[9 * (size_t)i + 3 * c + a] = (float)n[3 * (size_t)t[6 * i + 3 + c] + a];
				}
		DevMesh &dm = meshes[m];
		dm.node_first = (int32_t)nodes.size();
		dm.tri_first = (int32_t)tris.size();
		dm.num_triangles = nt;
		if (device_build(nt)) {
			// built on the GPU once the arrays are uploaded: reserve the mesh's part
			PendingBuild pb;
			pb.mesh = m;
			pb.pos.swap(pos);
			pb.nrm.swap(nrm);
			pending.push_back(std::move(pb));
			nodes.resize(nodes.size() + (size_t)(nt > 1 ? nt - 1 : 1));
			tris.resize(tris.size() + (size_t)nt);
			normals.resize(normals.size() + (size_t)nt);
			edges.resize(edges.size() + (size_t)nt);
			sc->mesh_build_seconds.push_back(0.0);
			continue;
		}
		BvhResult bvh;
		const auto build_t0 = std::chrono::steady_clock::now();
		const bool built = bvh_build_mesh(pos.data(), nt, bvh);
		sc->mesh_build_seconds.push_back(
				std::chrono::duration<double>(std::chrono::steady_clock::now() - build_t0).count());
		if (!built) {
			rtamd_set_error("mesh %d: could not build a BVH within the traversal stack depth", m);
			return RTAMD_ERR_LIMIT;
		}
		dm.root = bvh.root;
		nodes.insert(nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
		emit_leaf_ordered(pos.data(), nrm.data(), nt, bvh, tris, normals, edges);
	}
	if (nodes.empty())
		nodes.emplace_back();

	std::vector<float> texels;
	if (h.has_cubemap) {
		const rtamd_cubemap *bc = reinterpret_cast<const rtamd_cubemap *>(base + h.off_cubemap);
		for (int f = 0; f < 6; f++) {
			d.faces[f].width = bc->width[f];
			d.faces[f].height = bc->height[f];
			d.faces[f].offset = (int64_t)(texels.size() / 4);
			const float *src = reinterpret_cast<const float *>(base + bc->off_face[f]);
			texels.insert(texels.end(), src, src + 4 * (size_t)bc->width[f] * bc->height[f]);
		}
	}

	int rc;
	if ((rc = upload(sc, lights, &d.lights))) return rc;
	if ((rc = upload(sc, mats, &d.materials))) return rc;
	if ((rc = upload(sc, surfs, &d.surfaces))) return rc;
	if ((rc = upload(sc, meshes, &d.meshes))) return rc;
	if ((rc = upload(sc, nodes, &d.nodes))) return rc;
	if ((rc = upload(sc, tris, &d.tris))) return rc;
	if ((rc = upload(sc, normals, &d.normals))) return rc;
	if ((rc = upload(sc, edges, &d.edges))) return rc;
	if ((rc = upload(sc, texels, &d.texels))) return rc;

	// RTAMD_BVH_MAX_DEPTH (test knob): device-built trees deeper than this are replaced
	// by the host's tree, as those deeper than the traversal stack always are
	int depth_limit = RT_MAX_STACK - 2;
	if (const char *dl = getenv("RTAMD_BVH_MAX_DEPTH")) {
		const int v = atoi(dl);
		if (v >= 1 && v < depth_limit)
			depth_limit = v;
	}
	sc->mesh_built_on_device.assign(h.num_meshes, 0);
	for (const PendingBuild &pb : pending) {
		const auto build_t0 = std::chrono::steady_clock::now();
		DevMesh &dm = meshes[pb.mesh];
		const int nt = dm.num_triangles;
		// staging copies of the corners, only read by the build: freed as soon as it returns
		float *d_pos = nullptr, *d_nrm = nullptr;
		const size_t bytes = (size_t)nt * 9 * sizeof(float);
		hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_pos), bytes);
		if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_nrm), bytes);
		if (e == hipSuccess) e = hipMemcpy(d_pos, pb.pos.data(), bytes, hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(d_nrm, pb.nrm.data(), bytes, hipMemcpyHostToDevice);
		int root = 0, depth = 0;
		if (e == hipSuccess)
			e = bvh_build_device(d_pos, d_nrm, nt, const_cast<DevBvhNode *>(d.nodes) + dm.node_first,
					const_cast<DevTriangle *>(d.tris) + dm.tri_first, const_cast<DevTriangle *>(d.normals) + dm.tri_first,
					const_cast<DevTriEdges *>(d.edges) + dm.tri_first, &root, &depth, nullptr);
		if (d_pos) (void)hipFree(d_pos);
		if (d_nrm) (void)hipFree(d_nrm);
		if (e != hipSuccess) {
			rtamd_set_error("mesh %d: BVH build on the device failed: %s", pb.mesh, hipGetErrorString(e));
			return RTAMD_ERR_HIP;
		}
		if (depth > depth_limit) {
			// A radix tree over Morton codes can be deeper than the traversal stack (long runs
			// of equal or nearly equal codes); the host's builder bounds its depth by falling
			// back to median splits.  The mesh's reserved ranges are overwritten with its tree.
			BvhResult bvh;
			if (!bvh_build_mesh(pb.pos.data(), nt, bvh)) {
				rtamd_set_error("mesh %d: could not build a BVH within the traversal stack depth", pb.mesh);
				return RTAMD_ERR_LIMIT;
			}
			std::vector<DevTriangle> htris, hnormals;
			std::vector<DevTriEdges> hedges;
			emit_leaf_ordered(pb.pos.data(), pb.nrm.data(), nt, bvh, htris, hnormals, hedges);
			if (bvh.nodes.size() > (size_t)(nt > 1 ? nt - 1 : 1)) {
				rtamd_set_error("mesh %d: internal error: host tree has %zu nodes for %d triangles", pb.mesh,
						bvh.nodes.size(), nt);
				return RTAMD_ERR_LIMIT;
			}
			if (!bvh.nodes.empty())
				HIP_TRY(hipMemcpy(const_cast<DevBvhNode *>(d.nodes) + dm.node_first, bvh.nodes.data(),
						bvh.nodes.size() * sizeof(DevBvhNode), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(const_cast<DevTriangle *>(d.tris) + dm.tri_first, htris.data(),
					htris.size() * sizeof(DevTriangle), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(const_cast<DevTriangle *>(d.normals) + dm.tri_first, hnormals.data(),
					hnormals.size() * sizeof(DevTriangle), hipMemcpyHostToDevice));
			HIP_TRY(hipMemcpy(const_cast<DevTriEdges *>(d.edges) + dm.tri_first, hedges.data(),
					hedges.size() * sizeof(DevTriEdges), hipMemcpyHostToDevice));
			root = bvh.root;
		} else
			sc->mesh_built_on_device[pb.mesh] = 1;
		dm.root = root;
		sc->mesh_build_seconds[pb.mesh] =
				std::chrono::duration<double>(std::chrono::steady_clock::now() - build_t0).count();
	}
	if (!pending.empty()) {
		// the mesh table again, now with the roots of the device-built trees
		HIP_TRY(hipMemcpy(const_cast<DevMesh *>(d.meshes), meshes.data(), meshes.size() * sizeof(DevMesh),
				hipMemcpyHostToDevice));
	}

	void *p = nullptr;
	HIP_TRY(hipMalloc(&p, sizeof(DevCounters) * RT_COUNTER_COPIES));
	sc->d_counters = static_cast<DevCounters *>(p);
	HIP_TRY(hipMalloc(&p, sizeof(unsigned) * 128));
	sc->d_qcount = static_cast<unsigned *>(p);
	HIP_TRY(hipHostMalloc(&p, sizeof(unsigned long long), hipHostMallocMapped));
	sc->h_overflow = static_cast<unsigned long long *>(p);
	*sc->h_overflow = 0;
	HIP_TRY(hipHostGetDevicePointer(&p, sc->h_overflow, 0));
	sc->d_overflow = static_cast<unsigned long long *>(p);
	sc->device = g_device;
	{
		// Test knobs.  RTAMD_QUEUE_CAPACITY bounds the size of the ray queues (frames are
		// then rendered in more chunks).  RTAMD_QUEUE_OVERFLOW_AT makes the kernels treat
		// the queues as that short although the chunks are sized for the real capacity:
		// the only way to reach the overflow handling, which the chunk size otherwise
		// rules out.
		const char *cap = getenv("RTAMD_QUEUE_CAPACITY");
		const unsigned long long c = cap ? strtoull(cap, nullptr, 0) : 0;
		sc->queue_capacity_max = c >= RT_BLOCK && c <= QUEUE_CAPACITY_LIMIT ? (unsigned)c : QUEUE_CAPACITY_MAX;
		const char *at = getenv("RTAMD_QUEUE_OVERFLOW_AT");
		sc->overflow_test_capacity = at ? (unsigned)strtoul(at, nullptr, 0) : 0;
		// RTAMD_GLIBC_TABLE_DRAWS bounds the serial-stream table (frames are then rendered
		// in more groups of tile rows)
		const char *draws = getenv("RTAMD_GLIBC_TABLE_DRAWS");
		const unsigned long long dr = draws ? strtoull(draws, nullptr, 0) : 0;
		sc->glibc_capacity_max = dr >= 1024 && dr <= GLIBC_TABLE_DRAWS_MAX ? dr : GLIBC_TABLE_DRAWS_DEFAULT;
	}
	return 0;
}

extern "C" int rtamd_scene_create(const void *blob, size_t blob_size, rtamd_scene **scene)
{
	if (!scene) {
		rtamd_set_error("rtamd_scene_create: NULL output");
		return RTAMD_ERR_ARG;
	}
	*scene = nullptr;
	if (g_device < 0) {
		rtamd_set_error("rtamd_scene_create: rtamd_init has not been called");
		return RTAMD_ERR_ARG;
	}
	int rc = rtamd_blob_validate(blob, blob_size);
	if (rc)
		return rc;
	rtamd_scene *sc = new rtamd_scene();
	sc->device = g_device;
	if (hipSetDevice(g_device) != hipSuccess) {
		rtamd_set_error("rtamd_scene_create: cannot select device %d", g_device);
		delete sc;
		return RTAMD_ERR_HIP;
	}
	std::memcpy(&sc->header, blob, sizeof sc->header);
	if (sc->header.max_reflections > 60) {
		rtamd_set_error("max_reflections %d exceeds the limit of 60", sc->header.max_reflections);
		delete sc;
		return RTAMD_ERR_LIMIT;
	}
	rc = scene_create_impl(static_cast<const uint8_t *>(blob), sc);
	if (rc) {
		rtamd_scene_destroy(sc);
		return rc;
	}
	*scene = sc;
	return 0;
}

extern "C" int rtamd_scene_mesh_build_seconds(const rtamd_scene *sc, int mesh, double *seconds)
{
	if (!sc || !seconds || mesh < 0 || mesh >= (int)sc->mesh_build_seconds.size()) {
		rtamd_set_error("rtamd_scene_mesh_build_seconds: bad argument");
		return RTAMD_ERR_ARG;
	}
	*seconds = sc->mesh_build_seconds[mesh];
	return 0;
}

extern "C" int rtamd_scene_mesh_built_on_device(const rtamd_scene *sc, int mesh, int *on_device)
{
	if (!sc || !on_device || mesh < 0 || mesh >= (int)sc->mesh_build_seconds.size()) {
		rtamd_set_error("rtamd_scene_mesh_built_on_device: bad argument");
		return RTAMD_ERR_ARG;
	}
	*on_device = mesh < (int)sc->mesh_built_on_device.size() && sc->mesh_built_on_device[mesh] ? 1 : 0;
	return 0;
}

extern "C" int rtamd_scene_dims(const rtamd_scene *sc, int *width, int *height, int *spp)
{
	if (!sc) {
		rtamd_set_error("rtamd_scene_dims: NULL scene");
		return RTAMD_ERR_ARG;
	}
	if (width) *width = sc->dev.width;
	if (height) *height = sc->dev.height;
	if (spp) *spp = sc->dev.spp;
	return 0;
}

// Fills in defaults and checks a parameter block against the frame.
static int normalise_params(int width, int height, const rtamd_render_params *in, rtamd_render_params *p)
{
	if (!in) {
		rtamd_set_error("render parameters are NULL");
		return RTAMD_ERR_ARG;
	}
	*p = *in;
	if (p->tile_w == 0) p->tile_w = 32;
	if (p->tile_h == 0) p->tile_h = 32;
	if (p->tile_stride == 0) p->tile_stride = 1;
	if (p->tile_w < 1 || p->tile_h < 1 || p->tile_w > 4096 || p->tile_h > 4096 || p->tile_stride < 1 ||
			p->tile_first < 0 || p->tile_first >= p->tile_stride) {
		rtamd_set_error("bad tiling: tile %dx%d first %d stride %d", p->tile_w, p->tile_h, p->tile_first,
				p->tile_stride);
		return RTAMD_ERR_ARG;
	}
	bool whole = p->x0 == 0 && p->y0 == 0 && p->x1 == 0 && p->y1 == 0;
	if (!whole && (p->x0 < 0 || p->y0 < 0 || p->x1 > width || p->y1 > height || p->x0 >= p->x1 || p->y0 >= p->y1)) {
		rtamd_set_error("bad window [%d,%d)x[%d,%d) for a %dx%d frame", p->x0, p->x1, p->y0, p->y1, width, height);
		return RTAMD_ERR_ARG;
	}
	if (p->rng_mode != RTAMD_RNG_GLIBC && p->rng_mode != RTAMD_RNG_KEYED) {
		rtamd_set_error("unknown rng_mode %d", p->rng_mode);
		return RTAMD_ERR_ARG;
	}
	return 0;
}

// ---- the serial rand() stream, generated on the device ---------------------------
//
// Draw 2*((j*W+i)*aa^2+k)+{0,1} feeds sample k of pixel (i,j): the draws of a run of
// pixel rows are one contiguous run of the stream.  A render call walks its tile
// rows in groups whose draws fit a bounded table; k_glibc_fill produces each group's
// run by jumping into the stream (polynomial arithmetic, front/glibc_rand.c) right
// before the primary kernel that reads it, on the same stream.

// The tabulated powers x^(BLOCK * i) and x^(BLOCK * 1024 * i), i < 1024: the same for
// every seed and scene, computed once per process.
static const std::vector<uint32_t> &glibc_powers_host()
{
	static std::vector<uint32_t> table;
	if (table.empty()) {
		std::vector<uint32_t> t((size_t)2048 * RT_GLIBC_POLY);
		for (int level = 0; level < 2; level++) {
			uint32_t step[RT_GLIBC_POLY], cur[RT_GLIBC_POLY] = { 1 }, next[RT_GLIBC_POLY];
			front_glibc_poly_pow(level ? (uint64_t)RT_GLIBC_BLOCK * 1024 : (uint64_t)RT_GLIBC_BLOCK, step);
			for (int i = 0; i < 1024; i++) {
				std::memcpy(&t[((size_t)level * 1024 + i) * RT_GLIBC_POLY], cur, sizeof cur);
				front_glibc_poly_mul(cur, step, next);
				std::memcpy(cur, next, sizeof cur);
			}
		}
		table.swap(t);
	}
	return table;
}

// Uploads the powers (once) and the seed's window (when the seed changes) into the
// given device buffers.  No allocation.
static int glibc_upload_tables(rtamd_scene *sc, uint32_t seed, uint32_t *d_powers, bool powers_valid,
		uint32_t *d_window, bool window_valid, hipStream_t stream)
{
	if (!powers_valid) {
		const std::vector<uint32_t> &t = glibc_powers_host();
		HIP_TRY(hipMemcpyAsync(d_powers, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
	}
	if (!window_valid) {
		uint32_t w[RT_GLIBC_BLOCK + RT_GLIBC_POLY];
		front_glibc_window(seed, w, RT_GLIBC_BLOCK + RT_GLIBC_POLY);
		HIP_TRY(hipMemcpyAsync(d_window, w, sizeof w, hipMemcpyHostToDevice, stream));
		// the source is on this function's stack: the copy must have left it
		HIP_TRY(hipStreamSynchronize(stream));
	}
	(void)sc;
	return 0;
}

// Fills table[0 .. count) with the draws first .. first + count - 1.
static int glibc_fill(const uint32_t *d_powers, const uint32_t *d_window, int32_t *table,
		unsigned long long first, unsigned long long count, hipStream_t stream)
{
	GlibcJump start;
	front_glibc_poly_pow(RT_GLIBC_FIRST_DRAW + first, start.c);
	HIP_TRY(rt_launch_glibc_fill(start, d_powers, d_window, table, count, stream));
	return 0;
}

// rtamd_render / rtamd_camera_rays: the scene's own buffers, allocated on first use;
// the table holds at least `want` draws afterwards (or the configured maximum).
static int ensure_glibc_buffers(rtamd_scene *sc, uint32_t seed, unsigned long long want,
		unsigned long long at_least, hipStream_t stream)
{
	if (!sc->dev.antialiasing)
		return 0;
	void *p = nullptr;
	if (!sc->d_glibc_powers || !sc->d_glibc_window) {
		// both or neither: a failure of the second allocation must not leave a powers
		// table that the next call takes for uploaded
		void *pw = nullptr, *win = nullptr;
		hipError_t e = hipMalloc(&pw, (size_t)2048 * RT_GLIBC_POLY * sizeof(uint32_t));
		if (e == hipSuccess)
			e = hipMalloc(&win, (RT_GLIBC_BLOCK + RT_GLIBC_POLY) * sizeof(uint32_t));
		if (e != hipSuccess) {
			if (pw) (void)hipFree(pw);
			rtamd_set_error("allocating the rand() stream tables failed: %s", hipGetErrorString(e));
			return RTAMD_ERR_HIP;
		}
		sc->d_glibc_powers = static_cast<uint32_t *>(pw);
		sc->d_glibc_window = static_cast<uint32_t *>(win);
		sc->glibc_powers_uploaded = false;
		sc->glibc_window_valid = false;
	}
	const bool had_powers = sc->glibc_powers_uploaded;
	const bool window_ok = sc->glibc_window_valid && sc->glibc_window_seed == seed;
	if (!window_ok)
		HIP_TRY(hipStreamSynchronize(stream));     // an earlier fill may still read the old window
	int rc = glibc_upload_tables(sc, seed, sc->d_glibc_powers, had_powers, sc->d_glibc_window, window_ok, stream);
	if (rc)
		return rc;
	sc->glibc_powers_uploaded = true;
	sc->glibc_window_seed = seed;
	sc->glibc_window_valid = true;
	// the configured size, but never less than one row of tiles needs (if that is allowed at all)
	if (want > sc->glibc_capacity_max)
		want = sc->glibc_capacity_max;
	if (want < at_least && at_least <= GLIBC_TABLE_DRAWS_MAX)
		want = at_least;
	if (sc->glibc_capacity < want) {
		if (sc->d_glibc) {
			HIP_TRY(hipStreamSynchronize(stream));
			(void)hipFree(sc->d_glibc);
			sc->d_glibc = nullptr;
			sc->glibc_capacity = 0;
		}
		HIP_TRY(hipMalloc(&p, want * sizeof(int32_t)));
		sc->d_glibc = static_cast<int32_t *>(p);
		sc->glibc_capacity = want;
	}
	return 0;
}

// Device buffers one render call works in.  rtamd_render takes them from the
// scene (allocated on first use, grown when a larger frame comes);
// rtamd_render_noalloc carves them out of this device's static scratch array.
struct RenderWork {
	long long *accum;            // 3 x int64 per owned pixel, or null (direct frames)
	float *fsum;                 // progressive frames whose samples share a wave: float4 per pixel, else null
	QueueItem *queue[3];         // [0]: rays of depth 1; [1], [2]: deeper levels, alternating
	unsigned queue_capacity;
	unsigned *qcount;            // 128 counters, one per bounce level
	DevCounters *counters;
	BounceWork bounce;           // hits == null: unsorted bounce pass
	// serial-stream mode with antialiasing: the draw table and what fills it
	int32_t *glibc;
	unsigned long long glibc_capacity;
	const uint32_t *glibc_powers, *glibc_window;
};

// Owns the HIP events of one call so that every return path releases them.
struct EventSet {
	hipEvent_t whole[2] = { nullptr, nullptr };
	std::vector<hipEvent_t> chunks;
	~EventSet()
	{
		for (hipEvent_t e : chunks)
			(void)hipEventDestroy(e);
		for (hipEvent_t e : whole)
			if (e)
				(void)hipEventDestroy(e);
	}
};
const size_t MAX_TIMED_CHUNKS = 2048;

// Children one path vertex can spawn: reflection_samples at depths 0 and 1 of a
// glossy material, one otherwise (shading.c:88).
static unsigned long long vertex_fan(const rtamd_scene *sc)
{
	const DevScene &dev = sc->dev;
	if (!(dev.any_reflective && sc->glossy))
		return 1;
	return (unsigned)(dev.reflection_samples > 1 ? dev.reflection_samples : 1);
}

// Whether rays of depth 1 can fan out again: then their queue is worked off in
// slices whose children fit the next queue.
static bool second_fan(const rtamd_scene *sc)
{
	return sc->dev.any_reflective && sc->glossy && sc->dev.max_reflections >= 2;
}

// A queue overflow of an earlier asynchronous call: the resolve step of that call
// left the count in host-visible memory.
static int take_sticky_overflow(rtamd_scene *sc)
{
	if (sc->h_overflow && *sc->h_overflow) {
		rtamd_set_error("a ray queue overflowed in an earlier render of this scene (%llu rays dropped): "
				"that frame is incomplete", (unsigned long long)*sc->h_overflow);
		*sc->h_overflow = 0;
		return RTAMD_ERR_QUEUE;
	}
	return 0;
}

// Renders the pixels [pix_lo, pix_hi) of the owner's tile sequence (whole tiles: the
// run of owned tiles the call's window can touch).  w.accum holds one slot per pixel
// of that run; the kernels index accumulators by position in the whole sequence, so
// they get the pointer moved back by the run's start.
static unsigned long long pixels_of(unsigned long long lo, unsigned long long hi) { return hi - lo; }

static int render_impl(rtamd_scene *sc, const rtamd_render_params &p, const RenderWork &w, bool direct,
		unsigned long long pix_lo, unsigned long long pix_hi, void *d_colour, void *d_hits,
		rtamd_render_stats *stats, hipStream_t stream)
{
	DevScene dev = sc->dev;
	const bool progressive = (p.flags & RTAMD_RENDER_PROGRESSIVE) != 0;
	if (progressive) {
		dev.sub_stride = p.sample_stride > 0 ? p.sample_stride : 1;
		dev.sub_first = p.sample_first;
		dev.sub_count = (dev.spp - dev.sub_first + dev.sub_stride - 1) / dev.sub_stride;
		if (p.sample_count > 0 && p.sample_count < dev.sub_count)
			dev.sub_count = p.sample_count;
		sc->progress_samples += dev.sub_count;
		dev.resolve_samples = sc->progress_samples;
		if (w.fsum && (p.flags & RTAMD_RENDER_RESTART))
			HIP_TRY(hipMemsetAsync(w.fsum, 0, pixels_of(pix_lo, pix_hi) * 4 * sizeof(float), stream));
	}
	const unsigned long long pixels = pix_hi - pix_lo;
	const unsigned long long sample_lo = pix_lo * (unsigned)dev.sub_count, samples = pix_hi * (unsigned)dev.sub_count;
	long long *const accum = w.accum ? w.accum - 3 * (long long)pix_lo : nullptr;
	const unsigned long long fan = vertex_fan(sc);
	unsigned long long chunk = samples - sample_lo;
	if (!chunk)
		chunk = RT_BLOCK;
	if (dev.any_reflective) {
		chunk = w.queue_capacity / fan;
		chunk -= chunk % RT_BLOCK;
		if (chunk < RT_BLOCK) {
			rtamd_set_error("reflection_samples %d needs a larger ray queue than %u entries",
					dev.reflection_samples, w.queue_capacity);
			return RTAMD_ERR_LIMIT;
		}
	}
	// one launch covers at most 2^31 - 1 workgroups
	const unsigned long long max_chunk = 0x7fffffffull * RT_BLOCK;
	if (chunk > max_chunk)
		chunk = max_chunk - max_chunk % RT_BLOCK;

	// Serial-stream mode with antialiasing: the tile rows are walked in groups whose
	// draws fit the table; otherwise there is one group, all rows.
	int row0 = 0, row1 = 0;
	rt_window_tile_rows(dev.width, dev.height, p, &row0, &row1);
	const bool streamed = p.rng_mode == RTAMD_RNG_GLIBC && dev.antialiasing;
	int rows_per_group = row1 - row0;
	const unsigned long long draws_per_pixel_row = 2ull * (unsigned)dev.width * (unsigned)dev.spp;
	if (streamed) {
		const unsigned long long per_tile_row = draws_per_pixel_row * (unsigned)p.tile_h;
		if (!w.glibc || w.glibc_capacity < per_tile_row) {
			rtamd_set_error("the rand() stream table holds %llu draws; one row of %d-pixel-high tiles of this "
					"frame needs %llu (use smaller tiles, or RTAMD_RNG_KEYED)", w.glibc_capacity, p.tile_h,
					per_tile_row);
			return RTAMD_ERR_LIMIT;
		}
		const unsigned long long fit = w.glibc_capacity / per_tile_row;
		if (fit < (unsigned long long)rows_per_group)
			rows_per_group = (int)fit;
	}
	if (rows_per_group < 1)
		rows_per_group = 1;

	unsigned kernel_capacity = w.queue_capacity;
	if (sc->overflow_test_capacity && sc->overflow_test_capacity < kernel_capacity)
		kernel_capacity = sc->overflow_test_capacity;

	EventSet ev;
	if (stats) {
		HIP_TRY(hipEventCreate(&ev.whole[0]));
		HIP_TRY(hipEventCreate(&ev.whole[1]));
		HIP_TRY(hipEventRecord(ev.whole[0], stream));
	}
	if (!direct && (!progressive || (p.flags & RTAMD_RENDER_RESTART)))
		HIP_TRY(hipMemsetAsync(w.accum, 0, pixels * 3 * sizeof(long long), stream));
	HIP_TRY(hipMemsetAsync(w.counters, 0, sizeof(DevCounters) * RT_COUNTER_COPIES, stream));

	int launches_primary = 0, launches_bounce = 0;
	const int levels = dev.any_reflective ? dev.max_reflections : 0;
	unsigned bounce_grid = 0;
	if (levels) {
		unsigned long long blocks = ((unsigned long long)w.queue_capacity + RT_BLOCK - 1) / RT_BLOCK;
		bounce_grid = blocks > 2048 ? 2048u : (unsigned)blocks;
	}
	for (int g0 = row0; g0 < row1; g0 += rows_per_group) {
		const int g1 = g0 + rows_per_group < row1 ? g0 + rows_per_group : row1;
		unsigned long long tile_lo = 0, tile_hi = 0;
		rt_tile_rows_range(dev.width, dev.height, p, g0, g1, &tile_lo, &tile_hi);
		const unsigned long long per_tile = (unsigned long long)p.tile_w * (unsigned)p.tile_h * (unsigned)dev.spp;
		const unsigned long long group_lo = tile_lo * per_tile, group_hi = tile_hi * per_tile;
		GlibcTable table = { w.glibc, 0 };
		if (streamed && group_hi > group_lo) {
			const unsigned long long y0 = (unsigned long long)g0 * (unsigned)p.tile_h;
			unsigned long long y1 = (unsigned long long)g1 * (unsigned)p.tile_h;
			if (y1 > (unsigned)dev.height)
				y1 = (unsigned)dev.height;
			table.first = y0 * draws_per_pixel_row;
			int rc = glibc_fill(w.glibc_powers, w.glibc_window, w.glibc, table.first, (y1 - y0) * draws_per_pixel_row,
					stream);
			if (rc)
				return rc;
		}
		for (unsigned long long first = group_lo; first < group_hi; first += chunk) {
			unsigned long long count = group_hi - first < chunk ? group_hi - first : chunk;
			if (levels)
				HIP_TRY(hipMemsetAsync(w.qcount, 0, sizeof(unsigned) * 128, stream));
			const bool timed = stats && ev.chunks.size() < 2 * MAX_TIMED_CHUNKS;
			// without statistics: the event pairs stay with the scene and are read later,
			// so the call itself never waits for the device
			const bool deferred = !stats && (p.flags & RTAMD_RENDER_TIME_KERNELS) &&
					sc->deferred_events.size() < 2 * MAX_TIMED_CHUNKS;
			if (timed || deferred) {
				hipEvent_t e = nullptr;
				HIP_TRY(hipEventCreate(&e));
				(timed ? ev.chunks : sc->deferred_events).push_back(e);
				HIP_TRY(hipEventRecord(e, stream));
			}
			HIP_TRY(rt_launch_primary(dev, p, first, count, table, accum,
					direct ? static_cast<float *>(d_colour) : nullptr, w.fsum, static_cast<rtamd_hit_record *>(d_hits),
					levels ? w.queue[0] : nullptr, levels ? w.qcount : nullptr, kernel_capacity, w.counters, stream));
			if (timed || deferred) {
				hipEvent_t e = nullptr;
				HIP_TRY(hipEventCreate(&e));
				(timed ? ev.chunks : sc->deferred_events).push_back(e);
				HIP_TRY(hipEventRecord(e, stream));
			}
			launches_primary++;
			// Rays of depth 1 wait in queue 0.  Where they can fan out again they are taken
			// in slices of capacity / fan entries, so that a slice's children fit a queue;
			// every slice is followed down to the last level before the next one starts
			// (slices beyond the population find nothing and cost a few empty launches).
			const unsigned slices = second_fan(sc) ? (unsigned)fan : 1u;
			const unsigned slice_len = w.queue_capacity / slices;
			for (unsigned sl = 0; sl < slices; sl++) {
				if (sl > 0 || levels > 1)
					HIP_TRY(hipMemsetAsync(w.qcount + 1, 0, sizeof(unsigned) * 127, stream));
				for (int level = 1; level <= levels; level++) {
					QueueSlice in;
					in.items = level == 1 ? w.queue[0] : w.queue[1 + (level & 1)];
					in.count = w.qcount + (level - 1);
					in.capacity = kernel_capacity;
					in.first = level == 1 ? sl * slice_len : 0;
					in.limit = level == 1 ? slice_len : kernel_capacity;
					QueueItem *out = level < levels ? w.queue[1 + ((level + 1) & 1)] : nullptr;
					if (w.bounce.hits)
						HIP_TRY(rt_launch_bounce_sorted(dev, in, accum, out, w.qcount + level, w.counters, w.bounce,
								bounce_grid, stream));
					else
						HIP_TRY(rt_launch_bounce(dev, in, accum, out, w.qcount + level, w.counters, bounce_grid,
								sc->bounce_packets, stream));
					launches_bounce++;
				}
			}
		}
	}
	HIP_TRY(rt_launch_collect_stats(w.counters, stream));
	if (!direct)
		HIP_TRY(rt_launch_resolve(dev, p, pix_lo, pixels, (w.fsum && !dev.any_reflective) ? nullptr : accum, w.fsum,
				static_cast<float *>(d_colour), w.counters,
				sc->d_overflow, stream));

	if (stats) {
		std::memset(stats, 0, sizeof *stats);
		HIP_TRY(hipEventRecord(ev.whole[1], stream));
		HIP_TRY(hipEventSynchronize(ev.whole[1]));
		HIP_TRY(hipEventElapsedTime(&stats->ms_total, ev.whole[0], ev.whole[1]));
		float sum = 0.0f;
		for (size_t i = 0; i + 1 < ev.chunks.size(); i += 2) {
			float ms = 0.0f;
			HIP_TRY(hipEventElapsedTime(&ms, ev.chunks[i], ev.chunks[i + 1]));
			sum += ms;
		}
		// frames with more chunks than are timed: scaled from the timed ones
		if ((size_t)launches_primary * 2 > ev.chunks.size() && !ev.chunks.empty())
			sum *= (float)launches_primary / (float)(ev.chunks.size() / 2);
		stats->ms_primary = sum;
		DevCounters c;
		HIP_TRY(hipMemcpy(&c, w.counters, sizeof c, hipMemcpyDeviceToHost));
		stats->rays_primary = c.rays_primary;
		stats->rays_shadow = c.rays_shadow;
		if (sc->unlit_point_lights)
			stats->rays_shadow += (unsigned long long)sc->unlit_point_lights * (c.rays_shadow / (unsigned)sc->kept_shadow_rays_per_hit);
		stats->rays_secondary = c.rays_secondary;
		stats->env_lookups = c.env_lookups;
		stats->queue_peak = c.queue_peak;
		stats->bvh_nodes = c.bvh_nodes;
		stats->tri_tests = c.tri_tests;
		stats->samples = samples - sample_lo;
		stats->launches_primary = launches_primary;
		stats->launches_bounce = launches_bounce;
		if (c.queue_overflow) {
			if (sc->h_overflow)
				*sc->h_overflow = 0;      // reported here, not again by the next call
			rtamd_set_error("a ray queue overflowed (%llu rays dropped)", c.queue_overflow);
			return RTAMD_ERR_QUEUE;
		}
	}
	return 0;
}

// Argument checks shared by the two render entry points.
// One process may hold scenes on several GPUs (rtamd_init once per device): every
// entry point that launches work makes the scene's device current first.
static int use_scene_device(const rtamd_scene *sc)
{
	int current = -1;
	HIP_TRY(hipGetDevice(&current));
	if (current != sc->device)
		HIP_TRY(hipSetDevice(sc->device));
	return 0;
}

static int render_prepare(rtamd_scene *sc, const rtamd_render_params *params, void *d_colour, void *d_hits,
		rtamd_render_params *p, unsigned long long *pix_lo, unsigned long long *pix_hi, bool *direct)
{
	if (!sc || !d_colour) {
		rtamd_set_error("rtamd_render: NULL scene or output");
		return RTAMD_ERR_ARG;
	}
	if (int rc0 = use_scene_device(sc))
		return rc0;
	int rc = take_sticky_overflow(sc);
	if (rc)
		return rc;
	if ((rc = normalise_params(sc->dev.width, sc->dev.height, params, p)))
		return rc;
	if (p->rng_mode == RTAMD_RNG_GLIBC && !sc->fixed_consumption) {
		rtamd_set_error("RTAMD_RNG_GLIBC needs a scene whose rand() consumption is fixed (no area "
				"light, no glossy material); use RTAMD_RNG_KEYED");
		return RTAMD_ERR_ARG;
	}
	if (d_hits && p->tile_stride != 1) {
		rtamd_set_error("hit records are only available for single-owner renders (tile_stride 1)");
		return RTAMD_ERR_ARG;
	}
	if (rt_owned_pixels(sc->dev.width, sc->dev.height, *p) > 0xffffffffull) {
		rtamd_set_error("too many pixels per call");
		return RTAMD_ERR_LIMIT;
	}
	unsigned long long tile_lo, tile_hi;
	int row0 = 0, row1 = 0;
	rt_window_tile_rows(sc->dev.width, sc->dev.height, *p, &row0, &row1);
	rt_tile_rows_range(sc->dev.width, sc->dev.height, *p, row0, row1, &tile_lo, &tile_hi);
	const unsigned long long per_tile = (unsigned long long)p->tile_w * (unsigned)p->tile_h;
	*pix_lo = tile_lo * per_tile;
	*pix_hi = tile_hi * per_tile;
	// Without reflections, and with all samples of a pixel inside one wave, the
	// primary kernel writes finished pixels itself: no accumulators, no resolve.
	*direct = !sc->dev.any_reflective && 64 % sc->dev.spp == 0;
	if (p->flags & RTAMD_RENDER_PROGRESSIVE) {
		const bool whole = (p->x0 == 0 && p->y0 == 0 && p->x1 == 0 && p->y1 == 0) ||
				(p->x0 == 0 && p->y0 == 0 && p->x1 == sc->dev.width && p->y1 == sc->dev.height);
		const int stride = p->sample_stride > 0 ? p->sample_stride : 1;
		if (!whole || p->tile_stride != 1 || d_hits) {
			rtamd_set_error("progressive renders take the whole frame, all tiles, no hit records");
			return RTAMD_ERR_ARG;
		}
		if (p->flags & RTAMD_RENDER_RESTART)
			sc->progress_samples = 0;
		const int spp = sc->dev.spp;
		if (p->sample_first < 0 || p->sample_first >= spp || stride > spp || p->sample_count < 0) {
			rtamd_set_error("bad sample subset: first %d stride %d count %d of %d samples per pixel", p->sample_first,
					p->sample_stride, p->sample_count, spp);
			return RTAMD_ERR_ARG;
		}
		if (64 % spp == 0) {
			// a pixel's samples are added in float, in order: the passes must follow that order
			int count = (spp - p->sample_first + stride - 1) / stride;
			if (p->sample_count > 0 && p->sample_count < count)
				count = p->sample_count;
			if (stride != 1 || p->sample_first != sc->progress_samples || 64 % count != 0) {
				rtamd_set_error("with %d samples per pixel a progressive frame takes its samples in order: "
						"sample_stride 1, sample_first = samples rendered so far (%d), sample_count a divisor of 64",
						spp, sc->progress_samples);
				return RTAMD_ERR_ARG;
			}
		}
		*direct = false;       // the sums live in the scene's buffers from pass to pass
	} else if (p->sample_first != 0 || p->sample_stride > 1) {
		rtamd_set_error("a sample subset needs RTAMD_RENDER_PROGRESSIVE");
		return RTAMD_ERR_ARG;
	}
	return 0;
}

extern "C" int rtamd_render(rtamd_scene *sc, const rtamd_render_params *params, void *d_colour, void *d_hits,
		rtamd_render_stats *stats, void *stream_v)
{
	hipStream_t stream = static_cast<hipStream_t>(stream_v);
	rtamd_render_params p;
	unsigned long long pix_lo = 0, pix_hi = 0;
	bool direct = false;
	int rc = render_prepare(sc, params, d_colour, d_hits, &p, &pix_lo, &pix_hi, &direct);
	if (rc)
		return rc;
	const unsigned long long pixels = pix_hi - pix_lo;
	const DevScene &dev = sc->dev;
	if (p.rng_mode == RTAMD_RNG_GLIBC) {
		// the draws of the tile rows this call renders, capped at the table's maximum
		int r0 = 0, r1 = 0;
		rt_window_tile_rows(dev.width, dev.height, p, &r0, &r1);
		const unsigned long long rows = (unsigned long long)(r1 - r0) * (unsigned)p.tile_h;
		const unsigned long long per_row = 2ull * (unsigned)dev.width * (unsigned)dev.spp;
		if ((rc = ensure_glibc_buffers(sc, p.seed, per_row * rows, per_row * (unsigned)p.tile_h, stream)))
			return rc;
	}
	const unsigned long long samples = pixels * (unsigned)dev.spp;
	// work buffers
	if (!direct && sc->accum_slots < pixels) {
		if (sc->d_accum) {
			HIP_TRY(hipStreamSynchronize(stream));
			(void)hipFree(sc->d_accum);
			sc->d_accum = nullptr;
			sc->accum_slots = 0;
		}
		void *ptr = nullptr;
		HIP_TRY(hipMalloc(&ptr, pixels * 3 * sizeof(long long) + 16));
		sc->d_accum = static_cast<long long *>(ptr);
		sc->accum_slots = pixels;
	}
	if (dev.any_reflective) {
		unsigned long long want = samples * vertex_fan(sc);
		unsigned cap = want > sc->queue_capacity_max ? sc->queue_capacity_max
				: (want < QUEUE_CAPACITY_MIN ? QUEUE_CAPACITY_MIN : (unsigned)want);
		const bool want_sorted = sc->sorted_bounce;
		if (sc->queue_capacity < cap || (want_sorted && sc->bounce_capacity < sc->queue_capacity)) {
			HIP_TRY(hipStreamSynchronize(stream));
			// A capacity the device cannot give (other processes share it, or the frame asks for
			// the limit) is halved until the buffers fit: smaller queues only mean more chunks.
			const unsigned floor_cap = (unsigned)(vertex_fan(sc) * RT_BLOCK > QUEUE_CAPACITY_MIN ? vertex_fan(sc) * RT_BLOCK
					: QUEUE_CAPACITY_MIN);
			unsigned try_cap = cap > sc->queue_capacity ? cap : sc->queue_capacity;
			for (;;) {
				for (int q = 0; q < 3; q++) {
					if (sc->d_queue[q])
						(void)hipFree(sc->d_queue[q]);
					sc->d_queue[q] = nullptr;
				}
				sc->queue_capacity = 0;
				free_bounce_work(sc);
				hipError_t e = hipSuccess;
				void *ptr = nullptr;
				for (int q = 0; q < 3 && e == hipSuccess; q++) {
					e = hipMalloc(&ptr, (size_t)try_cap * sizeof(QueueItem));
					if (e == hipSuccess)
						sc->d_queue[q] = static_cast<QueueItem *>(ptr);
				}
				if (e == hipSuccess && want_sorted) {
					const size_t n = try_cap;
					BounceWork &bw = sc->bounce;
					auto take = [&](size_t bytes, void **out) {
						if (e == hipSuccess && (e = hipMalloc(&ptr, bytes)) == hipSuccess)
							*out = ptr;
					};
					take(n * sizeof(BounceHit), reinterpret_cast<void **>(&bw.hits));
					for (int k = 0; k < 2; k++) {
						take(n * sizeof(unsigned), reinterpret_cast<void **>(&bw.keys[k]));
						take(n * sizeof(unsigned), reinterpret_cast<void **>(&bw.index[k]));
					}
					take((size_t)RT_SORT_RADIX * bw.sort_chunks_max * sizeof(unsigned), reinterpret_cast<void **>(&bw.counts));
					take(RT_SORT_RADIX * sizeof(unsigned), reinterpret_cast<void **>(&bw.totals));
					take(8 * sizeof(unsigned), reinterpret_cast<void **>(&bw.bounds));
				}
				if (e == hipSuccess) {
					sc->queue_capacity = try_cap;
					if (want_sorted)
						sc->bounce_capacity = try_cap;
					break;
				}
				(void)hipGetLastError();      // the failed allocation is dealt with here
				if (e != hipErrorOutOfMemory || try_cap / 2 < floor_cap) {
					rtamd_set_error("allocating ray queues of %u entries failed: %s", try_cap, hipGetErrorString(e));
					return RTAMD_ERR_HIP;
				}
				try_cap /= 2;
				try_cap -= try_cap % RT_BLOCK;
			}
		}
	}
	float *fsum = nullptr;
	if ((p.flags & RTAMD_RENDER_PROGRESSIVE) && 64 % dev.spp == 0) {
		if (sc->fsum_slots < pixels) {
			if (sc->d_fsum) {
				HIP_TRY(hipStreamSynchronize(stream));
				(void)hipFree(sc->d_fsum);
				sc->d_fsum = nullptr;
				sc->fsum_slots = 0;
			}
			void *ptr = nullptr;
			HIP_TRY(hipMalloc(&ptr, pixels * 4 * sizeof(float) + 16));
			HIP_TRY(hipMemsetAsync(ptr, 0, pixels * 4 * sizeof(float), stream));
			sc->d_fsum = static_cast<float *>(ptr);
			sc->fsum_slots = pixels;
		}
		fsum = sc->d_fsum;
	}
	RenderWork w;
	w.fsum = fsum;
	w.accum = direct ? nullptr : sc->d_accum;
	w.queue[0] = sc->d_queue[0];
	w.queue[1] = sc->d_queue[1];
	w.queue[2] = sc->d_queue[2];
	w.queue_capacity = sc->queue_capacity;
	w.qcount = sc->d_qcount;
	w.counters = sc->d_counters;
	w.bounce = sc->bounce;
	if (!sc->sorted_bounce)
		w.bounce.hits = nullptr;
	w.glibc = sc->d_glibc;
	w.glibc_capacity = sc->glibc_capacity;
	w.glibc_powers = sc->d_glibc_powers;
	w.glibc_window = sc->d_glibc_window;
	return render_impl(sc, p, w, direct, pix_lo, pix_hi, d_colour, d_hits, stats, stream);
}

// ---- the entry point that allocates nothing ------------------------------------
//
// One array per GPU, present from the moment the library's code object is loaded
// until the process ends.  Host code never uses the array's name as an address:
// rtamd_init asks HIP for the array's device address on the device it selects and
// keeps it in g_scratch_ptr.
#define RTAMD_NOALLOC_SCRATCH_BYTES (128ull << 20)
static __device__ __align__(256) unsigned char scratch[RTAMD_NOALLOC_SCRATCH_BYTES];

namespace {
const int MAX_DEVICES = 64;
unsigned char *g_scratch_ptr[MAX_DEVICES];
// where in a device's scratch the serial-stream tables of the last call lie, and for which seed
struct ScratchStream {
	uint32_t *powers = nullptr;
	uint32_t window_seed = 0;
	bool window_valid = false;
};
ScratchStream g_scratch_stream[MAX_DEVICES];

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}

// Called by rtamd_init with `device` current.
static int scratch_init(int device)
{
	if (device < 0 || device >= MAX_DEVICES) {
		rtamd_set_error("device %d is beyond the %d devices this library keeps scratch addresses for", device,
				MAX_DEVICES);
		return RTAMD_ERR_LIMIT;
	}
	if (g_scratch_ptr[device])
		return 0;
	void *ptr = nullptr;
	HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
	g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
	return 0;
}

extern "C" int rtamd_render_noalloc(rtamd_scene *sc, const rtamd_render_params *params, void *d_colour,
		void *stream_v)
{
	hipStream_t stream = static_cast<hipStream_t>(stream_v);
	rtamd_render_params p;
	unsigned long long pix_lo = 0, pix_hi = 0;
	bool direct = false;
	int rc = render_prepare(sc, params, d_colour, nullptr, &p, &pix_lo, &pix_hi, &direct);
	if (rc)
		return rc;
	const unsigned long long pixels = pix_hi - pix_lo;
	if (p.flags & RTAMD_RENDER_PROGRESSIVE) {
		rtamd_set_error("rtamd_render_noalloc: progressive frames keep their sums in the scene's own buffers; "
				"use rtamd_render");
		return RTAMD_ERR_ARG;
	}
	if (!pixels)
		return 0;          // an owner without tiles in the window: nothing to render
	p.flags &= ~RTAMD_RENDER_TIME_KERNELS;     // event pairs are host allocations: not from this entry point
	unsigned char *base = (sc->device >= 0 && sc->device < MAX_DEVICES) ? g_scratch_ptr[sc->device] : nullptr;
	if (!base) {
		rtamd_set_error("rtamd_render_noalloc: no scratch address for device %d (rtamd_init not called?)",
				sc->device);
		return RTAMD_ERR_ARG;
	}
	// Layout: counters | queue counters | hit-point bounds | digit totals | digit counts |
	// accumulators | queues 0 1 2 | hits | sort keys and indices (two of each).
	const bool sorted = sc->sorted_bounce && sc->dev.any_reflective;
	const unsigned chunks_max = 256;      // wave-chunks of the hit sort: 1 MiB of digit counts
	size_t off = 0;
	RenderWork w;
	std::memset(&w, 0, sizeof w);
	w.counters = reinterpret_cast<DevCounters *>(base + off);  off = align_up(off + sizeof(DevCounters) * RT_COUNTER_COPIES, 256);
	w.qcount = reinterpret_cast<unsigned *>(base + off);       off = align_up(off + 128 * sizeof(unsigned), 256);
	w.bounce.bounds = reinterpret_cast<unsigned *>(base + off); off = align_up(off + 8 * sizeof(unsigned), 256);
	w.bounce.totals = reinterpret_cast<unsigned *>(base + off);  off = align_up(off + RT_SORT_RADIX * sizeof(unsigned), 256);
	w.bounce.bits = sc->bounce.bits;
	w.bounce.small_below = sc->bounce.small_below;
	w.bounce.hits_per_wave = sc->bounce.hits_per_wave;
	w.bounce.packets_from = sc->bounce.packets_from;
	w.bounce.sort_chunks_max = chunks_max;
	if (sorted) {
		w.bounce.counts = reinterpret_cast<unsigned *>(base + off);
		off = align_up(off + (size_t)RT_SORT_RADIX * chunks_max * sizeof(unsigned), 256);
	}
	const bool streamed = p.rng_mode == RTAMD_RNG_GLIBC && sc->dev.antialiasing;
	if (streamed) {
		// jump polynomials and the seed's window keep their place from call to call (they
		// are uploaded again only when the seed changes); the table takes at most 32 MiB
		ScratchStream &st = g_scratch_stream[sc->device];
		uint32_t *powers = reinterpret_cast<uint32_t *>(base + off);
		off = align_up(off + (size_t)2048 * RT_GLIBC_POLY * sizeof(uint32_t), 256);
		uint32_t *window = reinterpret_cast<uint32_t *>(base + off);
		off = align_up(off + (RT_GLIBC_BLOCK + RT_GLIBC_POLY) * sizeof(uint32_t), 256);
		const bool same_place = st.powers == powers;
		const bool window_ok = same_place && st.window_valid && st.window_seed == p.seed;
		if (!window_ok) {
			// The tables are uploaded from host memory with a stream synchronisation, which
			// a stream under capture does not allow: the first call per seed (and after a
			// call with another layout) must be made outside capture.
			hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
			if (stream && hipStreamIsCapturing(stream, &capture) == hipSuccess &&
					capture != hipStreamCaptureStatusNone) {
				rtamd_set_error("rtamd_render_noalloc: the rand() stream tables for seed %u are not resident; "
						"make one call with this seed outside stream capture first", p.seed);
				return RTAMD_ERR_ARG;
			}
			HIP_TRY(hipStreamSynchronize(stream));
		}
		if ((rc = glibc_upload_tables(sc, p.seed, powers, same_place, window, window_ok, stream)))
			return rc;
		st.powers = powers;
		st.window_seed = p.seed;
		st.window_valid = true;
		int r0 = 0, r1 = 0;
		rt_window_tile_rows(sc->dev.width, sc->dev.height, p, &r0, &r1);
		unsigned long long draws = 2ull * (unsigned)sc->dev.width * (unsigned)sc->dev.spp *
				((unsigned long long)(r1 - r0) * (unsigned)p.tile_h);
		if (draws > (8ull << 20))
			draws = 8ull << 20;
		w.glibc = reinterpret_cast<int32_t *>(base + off);
		w.glibc_capacity = draws;
		w.glibc_powers = powers;
		w.glibc_window = window;
		off = align_up(off + (size_t)draws * sizeof(int32_t), 256);
	} else {
		// another layout follows: the stream tables lose their place
		g_scratch_stream[sc->device].powers = nullptr;
	}
	if (!direct) {
		const size_t bytes = (size_t)pixels * 3 * sizeof(long long);
		if (bytes > RTAMD_NOALLOC_SCRATCH_BYTES - off) {
			rtamd_set_error("rtamd_render_noalloc: %llu pixels of a reflective scene need %zu bytes of "
					"accumulators; the scratch array holds %llu bytes (render fewer tiles per call, or use "
					"rtamd_render)", pixels, bytes, RTAMD_NOALLOC_SCRATCH_BYTES);
			return RTAMD_ERR_LIMIT;
		}
		w.accum = reinterpret_cast<long long *>(base + off);
		off = align_up(off + bytes, 256);
	}
	if (sc->dev.any_reflective) {
		const size_t per_entry = 3 * sizeof(QueueItem) + (sorted ? sizeof(BounceHit) + 4 * sizeof(unsigned) : 0);
		size_t cap = (RTAMD_NOALLOC_SCRATCH_BYTES - off - 8 * 256) / per_entry;
		cap -= cap % 256;
		// as rtamd_render sizes its queues: what the call's samples can spawn, but never
		// less than the floor that keeps a chunk at a workgroup or more
		unsigned long long want = pixels * (unsigned)sc->dev.spp * vertex_fan(sc);
		if (want < QUEUE_CAPACITY_MIN)
			want = QUEUE_CAPACITY_MIN;
		if (cap > want)
			cap = (size_t)align_up((size_t)want, 256);
		if (cap > sc->queue_capacity_max)
			cap = sc->queue_capacity_max;
		if (cap < vertex_fan(sc) * RT_BLOCK) {
			rtamd_set_error("rtamd_render_noalloc: the scratch array leaves room for %zu queued rays, fewer "
					"than one workgroup of samples can spawn (%llu); use rtamd_render", cap,
					vertex_fan(sc) * RT_BLOCK);
			return RTAMD_ERR_LIMIT;
		}
		w.queue_capacity = (unsigned)cap;
		w.queue[0] = reinterpret_cast<QueueItem *>(base + off);  off = align_up(off + cap * sizeof(QueueItem), 256);
		w.queue[1] = reinterpret_cast<QueueItem *>(base + off);  off = align_up(off + cap * sizeof(QueueItem), 256);
		w.queue[2] = reinterpret_cast<QueueItem *>(base + off);  off = align_up(off + cap * sizeof(QueueItem), 256);
		if (sorted) {
			w.bounce.hits = reinterpret_cast<BounceHit *>(base + off);  off = align_up(off + cap * sizeof(BounceHit), 256);
			for (int k = 0; k < 2; k++) {
				w.bounce.keys[k] = reinterpret_cast<unsigned *>(base + off);   off = align_up(off + cap * sizeof(unsigned), 256);
				w.bounce.index[k] = reinterpret_cast<unsigned *>(base + off);  off = align_up(off + cap * sizeof(unsigned), 256);
			}
		}
	}
	if (off > RTAMD_NOALLOC_SCRATCH_BYTES) {
		rtamd_set_error("rtamd_render_noalloc: internal layout error (%zu bytes)", off);
		return RTAMD_ERR_LIMIT;
	}
	return render_impl(sc, p, w, direct, pix_lo, pix_hi, d_colour, nullptr, nullptr, stream);
}

extern "C" int rtamd_trace(rtamd_scene *sc, const void *d_rays, uint64_t n, void *d_hits, void *stream_v)
{
	if (!sc || (n && (!d_rays || !d_hits))) {
		rtamd_set_error("rtamd_trace: NULL argument");
		return RTAMD_ERR_ARG;
	}
	if (n > (1ull << 31) * RT_BLOCK) {
		rtamd_set_error("rtamd_trace: too many rays in one call");
		return RTAMD_ERR_LIMIT;
	}
	if (int rc0 = use_scene_device(sc))
		return rc0;
	HIP_TRY(rt_launch_trace(sc->dev, static_cast<const double *>(d_rays), n,
			static_cast<rtamd_hit_record *>(d_hits), static_cast<hipStream_t>(stream_v)));
	return 0;
}

extern "C" int rtamd_camera_rays(rtamd_scene *sc, const rtamd_render_params *params, void *d_rays, void *stream_v)
{
	if (!sc || !d_rays) {
		rtamd_set_error("rtamd_camera_rays: NULL argument");
		return RTAMD_ERR_ARG;
	}
	hipStream_t stream = static_cast<hipStream_t>(stream_v);
	rtamd_render_params p;
	int rc = normalise_params(sc->dev.width, sc->dev.height, params, &p);
	if (rc)
		return rc;
	if ((rc = use_scene_device(sc)))
		return rc;
	GlibcTable table = { nullptr, 0 };
	if (p.rng_mode == RTAMD_RNG_GLIBC && sc->dev.antialiasing) {
		// every sample of the frame is written by one launch: the whole frame's draws
		const unsigned long long draws = 2ull * (unsigned)sc->dev.width * (unsigned)sc->dev.height * (unsigned)sc->dev.spp;
		const unsigned long long keep = sc->glibc_capacity_max;
		if (draws > GLIBC_TABLE_DRAWS_MAX) {
			rtamd_set_error("rtamd_camera_rays: the frame draws %llu values of the rand() stream; this entry "
					"point handles %llu (render windows of it, or use RTAMD_RNG_KEYED)", draws, GLIBC_TABLE_DRAWS_MAX);
			return RTAMD_ERR_LIMIT;
		}
		sc->glibc_capacity_max = draws > keep ? draws : keep;
		rc = ensure_glibc_buffers(sc, p.seed, draws, draws, stream);
		sc->glibc_capacity_max = keep;
		if (rc || (rc = glibc_fill(sc->d_glibc_powers, sc->d_glibc_window, sc->d_glibc, 0, draws, stream)))
			return rc;
		table.draws = sc->d_glibc;
	}
	if (rt_owned_pixels(sc->dev.width, sc->dev.height, p) * (unsigned)sc->dev.spp > 0x7fffffffull * 256) {
		rtamd_set_error("rtamd_camera_rays: too many samples in one call");
		return RTAMD_ERR_LIMIT;
	}
	HIP_TRY(rt_launch_camera_rays(sc->dev, p, table, static_cast<double *>(d_rays), stream));
	return 0;
}

extern "C" int rtamd_tiles_owned(int width, int height, const rtamd_render_params *params)
{
	rtamd_render_params p;
	int rc = normalise_params(width, height, params, &p);
	if (rc)
		return rc;
	return rt_tiles_owned(width, height, p);
}

extern "C" int rtamd_tiles_pack(int width, int height, const rtamd_render_params *params, const void *d_frame,
		void *d_packed, void *stream_v)
{
	rtamd_render_params p;
	int rc = normalise_params(width, height, params, &p);
	if (rc)
		return rc;
	if (!d_frame || !d_packed) {
		rtamd_set_error("rtamd_tiles_pack: NULL buffer");
		return RTAMD_ERR_ARG;
	}
	HIP_TRY(rt_launch_tiles_copy(width, height, p, static_cast<float *>(const_cast<void *>(d_frame)),
			static_cast<float *>(d_packed), 0, static_cast<hipStream_t>(stream_v)));
	return 0;
}

extern "C" int rtamd_tiles_unpack(int width, int height, const rtamd_render_params *params, const void *d_packed,
		void *d_frame, void *stream_v)
{
	rtamd_render_params p;
	int rc = normalise_params(width, height, params, &p);
	if (rc)
		return rc;
	if (!d_frame || !d_packed) {
		rtamd_set_error("rtamd_tiles_unpack: NULL buffer");
		return RTAMD_ERR_ARG;
	}
	HIP_TRY(rt_launch_tiles_copy(width, height, p, static_cast<float *>(d_frame),
			static_cast<float *>(const_cast<void *>(d_packed)), 1, static_cast<hipStream_t>(stream_v)));
	return 0;
}

extern "C" int rtamd_tiles_unpack_all(int width, int height, const rtamd_render_params *params, int first_owner,
		const void *d_packed_all, uint64_t owner_stride_bytes, void *d_frame, void *stream_v)
{
	rtamd_render_params p;
	int rc = normalise_params(width, height, params, &p);
	if (rc)
		return rc;
	if (!d_frame || !d_packed_all || first_owner < 0 || first_owner > p.tile_stride || owner_stride_bytes % 16) {
		rtamd_set_error("rtamd_tiles_unpack_all: bad argument");
		return RTAMD_ERR_ARG;
	}
	p.tile_first = 0;
	const uint64_t need = (uint64_t)rt_tiles_owned(width, height, p) * (unsigned)(p.tile_w * p.tile_h) * 16;
	if (owner_stride_bytes < need) {
		rtamd_set_error("rtamd_tiles_unpack_all: owner stride of %llu bytes is less than the %llu bytes owner 0's "
				"tiles take", (unsigned long long)owner_stride_bytes, (unsigned long long)need);
		return RTAMD_ERR_ARG;
	}
	HIP_TRY(rt_launch_tiles_unpack_all(width, height, p, first_owner, static_cast<float *>(d_frame),
			static_cast<const float *>(d_packed_all), owner_stride_bytes / 16, static_cast<hipStream_t>(stream_v)));
	return 0;
}

extern "C" int rtamd_scene_take_kernel_times(rtamd_scene *sc, float *ms_primary, int *launches)
{
	if (!sc || !ms_primary || !launches) {
		rtamd_set_error("rtamd_scene_take_kernel_times: NULL argument");
		return RTAMD_ERR_ARG;
	}
	*ms_primary = 0.0f;
	*launches = 0;
	if (int rc0 = use_scene_device(sc))
		return rc0;
	int rc = 0;
	for (size_t i = 0; i + 1 < sc->deferred_events.size(); i += 2) {
		float ms = 0.0f;
		hipError_t e = hipEventSynchronize(sc->deferred_events[i + 1]);
		if (e == hipSuccess)
			e = hipEventElapsedTime(&ms, sc->deferred_events[i], sc->deferred_events[i + 1]);
		if (e != hipSuccess && !rc) {
			rtamd_set_error("rtamd_scene_take_kernel_times: %s", hipGetErrorString(e));
			rc = RTAMD_ERR_HIP;
		}
		*ms_primary += ms;
		*launches += 1;
	}
	for (hipEvent_t e : sc->deferred_events)
		(void)hipEventDestroy(e);
	sc->deferred_events.clear();
	return rc;
}
