// HIP kernels of the render path for gfx950 (wave64).
//
// Arithmetic contract: every quantity that decides WHICH surface or triangle a
// ray hits, and the hit distance itself, is computed with the reference's own
// mixture of double vectors and float scalars, one IEEE rounding per
// operation (this file is compiled with -ffp-contract=off; the only fused
// operations are the explicitly written __builtin_fmaf calls in the BVH box
// test, which is this implementation's own conservative structure and never
// produces a reported value).  Reference lines are cited per function.
//
// Execution model: one lane per primary sample.  A wave holds 64 consecutive
// samples, i.e. for 64 samples per pixel all lanes of a wave share a pixel and
// its rays are coherent in the BVH.  Shadow rays are traced in the lane that
// owns the hit (they never leave registers); reflection / glossy rays go to a
// bounce queue in HBM: lanes that spawn rays are compacted with a wave-wide
// ballot + prefix count and the wave reserves its slots with one atomic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scene_dev.h"
#include "kernels.h"

namespace {

// Light shafts (see shaft_build): compile-time switches and capacities.
#ifndef RT_SHAFT
#define RT_SHAFT 1
#endif
#ifndef RT_SHAFT_CAP
#define RT_SHAFT_CAP 24        // leaves per list; a fuller shaft falls back to the BVH walk
#endif
#ifndef RT_SHAFT_LISTS
#define RT_SHAFT_LISTS 4       // mesh surfaces (in traversal order) that get a list
#endif

#ifdef RT_TRAVERSAL_STATS
#define RT_STAT(x) x
#else
#define RT_STAT(x)
#endif

struct d3 { double x, y, z; };
struct c3 { float r, g, b; };

// Scene tables are read through the constant address space: a wave-uniform
// index then becomes a scalar load (one fetch per wave through the scalar
// cache) instead of 64 identical vector loads.
#define CONSTAS __attribute__((address_space(4)))
typedef const CONSTAS DevSurface CSurface;
typedef const CONSTAS DevLight CLight;
typedef const CONSTAS DevMaterial CMaterial;
typedef const CONSTAS DevMesh CMesh;
template <typename T> __device__ __forceinline__ const CONSTAS T *in_const(const T *p)
{
	return (const CONSTAS T *)(unsigned long long)p;
}

#define DEV __device__ __forceinline__

DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

#ifdef RT_TRAVERSAL_STATS
__device__ unsigned long long g_stat_nodes, g_stat_tris;
// per wave, light and surface: [0..CAP] leaves listed, [CAP+1] list overflow, [CAP+2] surface culled,
// [CAP+3] surface touched but not listed (analytic shape, no free list), [CAP+4] nodes visited by shafts
__device__ unsigned long long g_stat_shaft[RT_SHAFT_CAP + 8];
// wave-level event counts: see the EV_ names
__device__ unsigned long long g_stat_ev[16];
enum { EV_BBOX_PRIMARY, EV_BBOX_SHADOW, EV_PRIM_PRIMARY, EV_PRIM_SHADOW, EV_MESH_WALK_PRIMARY, EV_MESH_WALK_SHADOW,
	EV_MESH_LIST_SHADOW, EV_LIST_BOX, EV_TRI_STAGE2, EV_TRI_EXACT, EV_SHADE, EV_SHADOW_RAYS, EV_CULLED };
#define RT_EV(e) do { if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_stat_ev[e], 1ull); } while (0)
#else
#define RT_EV(e) do { } while (0)
#endif

// Build option RT_TIMING=1|2: per-wave cycle counts per code region (1: the
// stages of a sample, 2: also the inside of trace()), summed over the launch
// and printed by k_collect_stats.  The counter reads perturb what they measure
// (more so at level 2); the shares are a guide, not a measurement of the
// uninstrumented kernel.
#ifdef RT_TIMING
enum { TM_KERNEL, TM_SETUP, TM_TRACE_PRIMARY, TM_FINISH, TM_LIGHT_SETUP, TM_TRACE_SHADOW, TM_SHADE, TM_ENV,
	TM_SPAWN, TM_TAIL, TM_SHAFT, TM_BOUNDS,
	TM_IN_BBOX, TM_IN_XFORM, TM_IN_PRIM, TM_IN_MESH_SETUP, TM_IN_NODES, TM_IN_LEAVES, TM_IN_POP, TM_IN_SELECT,
	TM_SHADOW_OFFSET = 8, TM_COUNT = TM_IN_SELECT + 1 + TM_SHADOW_OFFSET };
__device__ unsigned long long g_timing[TM_COUNT];
__shared__ unsigned long long s_timing[RT_BLOCK / 64][TM_COUNT];
DEV unsigned long long tm_now() { return __builtin_readcyclecounter(); }
DEV void tm_add(int cat, unsigned long long t0)
{
	const unsigned long long dt = tm_now() - t0;
	if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
		atomicAdd(&s_timing[threadIdx.x >> 6][cat], dt);
}
#define TM_BEGIN(v) const unsigned long long v = tm_now()
#define TM_END(cat, v) tm_add(cat, v)
#if RT_TIMING >= 2
#define TM2_BEGIN(v) TM_BEGIN(v)
#define TM2_END(cat, v) tm_add((cat) + (ANYHIT ? TM_SHADOW_OFFSET : 0), v)
#else
#define TM2_BEGIN(v)
#define TM2_END(cat, v)
#endif
#else
#define TM_BEGIN(v)
#define TM_END(cat, v)
#define TM2_BEGIN(v)
#define TM2_END(cat, v)
#endif

// ---- vector.c ---------------------------------------------------------------
DEV d3 vadd(d3 a, d3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }          // vector.c:11
DEV d3 vsub(d3 a, d3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }          // vector.c:21
DEV double vdot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // vector.c:31
DEV double vlen(d3 a) { return __builtin_sqrt(vdot(a, a)); }                     // vector.c:36
DEV d3 vscale(double s, d3 a) { return { a.x * s, a.y * s, a.z * s }; }          // vector.c:41
DEV d3 vnorm(d3 a) { return vscale(1 / vlen(a), a); }                            // vector.c:51
DEV d3 vcross(d3 a, d3 b)                                                        // vector.c:56
{
	return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
template <typename P> DEV d3 ld3(P p) { return { p[0], p[1], p[2] }; }

// ---- colour.c ---------------------------------------------------------------
DEV c3 cadd(c3 a, c3 b) { return { a.r + b.r, a.g + b.g, a.b + b.b }; }          // colour.c:8
DEV c3 cmul(c3 a, c3 b) { return { a.r * b.r, a.g * b.g, a.b * b.b }; }          // colour.c:19
DEV c3 cscale(float s, c3 a) { return { s * a.r, s * a.g, s * a.b }; }           // colour.c:30
template <typename P> DEV c3 lc3(P p) { return { p[0], p[1], p[2] }; }

#define M4(m, i, j) (m)[4 * (j) + (i)]

// matrix.c:61-81 + vector.c:95-104.  For an affine matrix the homogeneous
// coordinate is exactly 1 and x / 1 == x, so the divide is skipped.
// world_to_model whose 3x3 part is exactly the identity (all zeros +0): the
// products with 0 and 1 only matter for the sign of an exact zero, so
//   point:     x' = x + T   unless x == 0 and T == 0
//   direction: x' = x       unless x == 0
// and those cases take the general formula.  Bit-identical to matrix.c:61-96.
DEV d3 xform_point_general(CSurface &s, d3 v);
DEV d3 xform_dir_general(const CONSTAS double *m, d3 v);

DEV d3 xform_point(CSurface &s, d3 v)
{
	if (s.linear_identity) {
		const double tx = s.w2m[12], ty = s.w2m[13], tz = s.w2m[14];
		if (!((v.x == 0 && tx == 0) || (v.y == 0 && ty == 0) || (v.z == 0 && tz == 0)))
			return { v.x + tx, v.y + ty, v.z + tz };
	}
	return xform_point_general(s, v);
}

DEV d3 xform_dir(CSurface &s, d3 v)
{
	if (s.linear_identity && v.x != 0 && v.y != 0 && v.z != 0)
		return v;
	return xform_dir_general(s.w2m, v);
}

DEV d3 xform_point_general(CSurface &s, d3 v)
{
	const CONSTAS double *m = s.w2m;
	const double v0 = v.x, v1 = v.y, v2 = v.z, v3 = 1.0;
	double x = M4(m, 0, 0) * v0 + M4(m, 0, 1) * v1 + M4(m, 0, 2) * v2 + M4(m, 0, 3) * v3;
	double y = M4(m, 1, 0) * v0 + M4(m, 1, 1) * v1 + M4(m, 1, 2) * v2 + M4(m, 1, 3) * v3;
	double z = M4(m, 2, 0) * v0 + M4(m, 2, 1) * v1 + M4(m, 2, 2) * v2 + M4(m, 2, 3) * v3;
	if (s.affine)
		return { x, y, z };
	double w = M4(m, 3, 0) * v0 + M4(m, 3, 1) * v1 + M4(m, 3, 2) * v2 + M4(m, 3, 3) * v3;
	return { x / w, y / w, z / w };
}

// matrix.c:83-96
DEV d3 xform_dir_general(const CONSTAS double *m, d3 v)
{
	const double v0 = v.x, v1 = v.y, v2 = v.z;
	d3 out;
	out.x = M4(m, 0, 0) * v0 + M4(m, 0, 1) * v1 + M4(m, 0, 2) * v2 + M4(m, 0, 3) * 0;
	out.y = M4(m, 1, 0) * v0 + M4(m, 1, 1) * v1 + M4(m, 1, 2) * v2 + M4(m, 1, 3) * 0;
	out.z = M4(m, 2, 0) * v0 + M4(m, 2, 1) * v1 + M4(m, 2, 2) * v2 + M4(m, 2, 3) * 0;
	return out;
}

// transpose(world_to_model) applied as a direction, ray.c:492-493,576
DEV d3 xform_normal(const CONSTAS double *w, d3 n)
{
	const double n0 = n.x, n1 = n.y, n2 = n.z;
	d3 out;
	out.x = M4(w, 0, 0) * n0 + M4(w, 1, 0) * n1 + M4(w, 2, 0) * n2 + M4(w, 3, 0) * 0;
	out.y = M4(w, 0, 1) * n0 + M4(w, 1, 1) * n1 + M4(w, 2, 1) * n2 + M4(w, 3, 1) * 0;
	out.z = M4(w, 0, 2) * n0 + M4(w, 1, 2) * n1 + M4(w, 2, 2) * n2 + M4(w, 3, 2) * 0;
	return out;
}

struct Ray {
	d3 o, d;
	float tnear, tfar;
};

// Candidate bookkeeping: the normal is only needed for the winning surface, so
// each intersector reports how to rebuild it instead of the vector itself.
enum NormalCode : int { NC_FROM_T = 0, NC_POS_Z = 1, NC_NEG_Z = 2, NC_PLANE_POS = 3, NC_PLANE_NEG = 4,
	NC_MESH = 5, NC_SPHERE_UNIT = 6 };

struct Best {
	int surface;      // -1: none
	int tri;          // leaf-order triangle index within the mesh (meshes only)
	int ncode;
	float t;          // Hit.t is a float-valued double (ray.c:485,574)
	float a, b, c;    // barycentrics (meshes only)
};

// ---- keyed RNG (same definition as the CPU checker) ---------------------------
DEV uint32_t mix32(uint32_t h)
{
	h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
	return h;
}
DEV uint32_t hash2(uint32_t a, uint32_t b)
{
	return mix32(a ^ (mix32(b) + 0x9e3779b9u + (a << 6) + (a >> 2)));
}
DEV uint32_t key0_of(uint32_t seed, uint32_t pixel, uint32_t sample) { return hash2(hash2(seed, pixel), sample); }
DEV uint32_t key_child(uint32_t key, uint32_t child) { return hash2(key ^ 0xa511e9b3u, child + 1u); }
// The keyed stand-in for drand() (ray.c:14-17): a 31-bit value scaled into [0, 1]
// by the double nearest to 1/RAND_MAX (a product, not a quotient: this sequence
// is this implementation's own and the CPU checker defines it the same way).
#define KEYED_SCALE 4.6566128752457969e-10
DEV double keyed_drand(uint32_t key, uint32_t purpose, uint32_t index)
{
	return (double)(hash2(hash2(key, purpose), index) >> 1) * KEYED_SCALE;
}
enum { PURPOSE_AA = 0, PURPOSE_LIGHT0 = 1, PURPOSE_GLOSSY = 16 };

// ---- ray.c:580-641 ------------------------------------------------------------
DEV bool surface_bbox_test(const CONSTAS float *bbox, const Ray &ray, float &bnear, float &bfar)
{
	float xmin = bbox[0], xmax = bbox[3];
	float ymin = bbox[1], ymax = bbox[4];
	float zmin = bbox[2], zmax = bbox[5];
	float txmin, txmax, tymin, tymax, tzmin, tzmax;
	float xd = (float)ray.d.x, yd = (float)ray.d.y, zd = (float)ray.d.z;
	float xe = (float)ray.o.x, ye = (float)ray.o.y, ze = (float)ray.o.z;
	float xa, ya, za;

	xa = (float)(1.0 / (double)xd);
	if (xa >= 0) { txmin = (xmin - xe) * xa; txmax = (xmax - xe) * xa; }
	else         { txmin = (xmax - xe) * xa; txmax = (xmin - xe) * xa; }
	ya = (float)(1.0 / (double)yd);
	if (ya >= 0) { tymin = (ymin - ye) * ya; tymax = (ymax - ye) * ya; }
	else         { tymin = (ymax - ye) * ya; tymax = (ymin - ye) * ya; }
	za = (float)(1.0 / (double)zd);
	if (za >= 0) { tzmin = (zmin - ze) * za; tzmax = (zmax - ze) * za; }
	else         { tzmin = (zmax - ze) * za; tzmax = (zmin - ze) * za; }

	if ((txmin > tymax) || (tymin > txmax)) return false;
	if ((tymin > tzmax) || (tzmin > tymax)) return false;
	if ((tzmin > txmax) || (txmin > tzmax)) return false;

	// MAX/MIN macros of cgmath.h:15-16, written out so NaNs propagate the same way
	float m0 = tzmin > ray.tnear ? tzmin : ray.tnear;
	float m1 = tymin > m0 ? tymin : m0;
	bnear = txmin > m1 ? txmin : m1;
	float n0 = tzmax < ray.tfar ? tzmax : ray.tfar;
	float n1 = tymax < n0 ? tymax : n0;
	bfar = txmax < n1 ? txmax : n1;
	return true;
}

// ---- primitives, model space --------------------------------------------------
// Each returns the number of candidates (ray.c:500-524) and fills ts/ncodes.

// ray.c:69-97
DEV int plane_intersect(CSurface &s, d3 o, d3 d, float ts[2], int nc[2])
{
	d3 n = ld3(s.pn), axn = ld3(s.paxn), bxn = ld3(s.pbxn);
	double ddn = vdot(d, n);
	float test_t = (float)(-vdot(o, n) / ddn);
	d3 pos = vadd(o, vscale((double)test_t, d));
	float det = s.pdet;
	float alpha = (float)(vdot(pos, bxn) / (double)det);
	float beta = (float)(-vdot(pos, axn) / (double)det);
	if (alpha < 0 || alpha > 1 || beta < 0 || beta > 1)
		return 0;
	ts[0] = test_t;
	nc[0] = ddn < 0 ? NC_PLANE_POS : NC_PLANE_NEG;
	return 1;
}

// ray.c:99-120
DEV int disk_intersect(CSurface &s, d3 o, d3 d, float tnear, float ts[2], int nc[2])
{
	if (d.z == 0)
		return 0;
	double tt = -o.z / d.z;
	double xx = o.x + tt * d.x;
	double yy = o.y + tt * d.y;
	float r2 = s.radius * s.radius;
	if (xx * xx + yy * yy > (double)r2)
		return 0;
	ts[0] = (float)tt;
	nc[0] = (o.z + (double)tnear * d.z >= 0) ? NC_POS_Z : NC_NEG_Z;
	return 1;
}

// ray.c:122-151
DEV int sphere_intersect(CSurface &s, d3 o, d3 d, float ts[2], int nc[2])
{
	float radius = s.radius;
	float vv = (float)vdot(o, o);
	float vd = (float)vdot(o, d);
	float dd = (float)vdot(d, d);
	float disc = vd * vd - dd * (vv - radius * radius);
	if (disc < 0)
		return 0;
	else if (disc == 0) {
		ts[0] = -vd / dd;
		nc[0] = NC_SPHERE_UNIT;
		return 1;
	}
	float sq = __builtin_sqrtf(disc);
	ts[0] = (-vd - sq) / dd;
	ts[1] = (-vd + sq) / dd;
	nc[0] = nc[1] = NC_FROM_T;
	return 2;
}

// ray.c:153-271
DEV int cylinder_intersect(CSurface &s, d3 o, d3 d, float ts[2], int nc[2])
{
	const float height = s.height, radius = s.radius;
	const bool capped = s.capped != 0;
	float a = (float)(d.x * d.x + d.y * d.y);
	float b = (float)(2 * (o.x * d.x + o.y * d.y));
	float c = (float)(o.x * o.x + o.y * o.y - (double)(radius * radius));
	float disc = b * b - 4 * a * c;
	if (disc < 0)
		return 0;
	float sq = __builtin_sqrtf(disc);
	ts[0] = (-b - sq) / (2 * a);
	ts[1] = (-b + sq) / (2 * a);
	nc[0] = nc[1] = NC_FROM_T;
	float z0 = (float)(o.z + (double)ts[0] * d.z);
	float z1 = (float)(o.z + (double)ts[1] * d.z);
	if ((z0 < 0 && z1 < 0) || (z0 > height && z1 > height))
		return 0;
	float tbot = (float)(-o.z / d.z);
	float ttop = (float)(((double)height - o.z) / d.z);
	if ((z0 < 0 && z1 > height) || (z0 > height && z1 < 0)) {
		if (!capped)
			return 0;
		ts[0] = tbot; nc[0] = NC_NEG_Z;
		ts[1] = ttop; nc[1] = NC_POS_Z;
	} else if (z0 >= 0 && z0 <= height && z1 > height) {
		if (capped) { ts[1] = ttop; nc[1] = NC_POS_Z; }
		else        { ts[1] = ts[0]; }
	} else if (z0 >= 0 && z0 <= height && z1 < 0) {
		if (capped) { ts[1] = tbot; nc[1] = NC_NEG_Z; }
		else        { ts[1] = ts[0]; }
	} else if (z1 >= 0 && z1 <= height && z0 > height) {
		if (capped) { ts[0] = ttop; nc[0] = NC_POS_Z; }
		else        { ts[0] = ts[1]; }
	} else if (z1 >= 0 && z1 <= height && z0 < 0) {
		if (capped) { ts[0] = tbot; nc[0] = NC_NEG_Z; }
		else        { ts[0] = ts[1]; }
	} else if (z0 >= 0 && z0 <= height && z1 >= 0 && z1 <= height) {
		// nothing to fix
	} else {
		return 0;   // the reference traps here (ray.c:263-268); reported as a miss
	}
	return 2;
}

// ray.c:273-324 (reports ONE candidate, ray.c:323)
DEV int cone_intersect(CSurface &s, d3 o, d3 d, float ts[2], int nc[2])
{
	float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
	float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
	float R = s.radius, h = s.height;
	float k = R / h;
	float a = dx * dx + dy * dy - (k * dz) * (k * dz);
	float b = 2 * (ox * dx + oy * dy + (k * k) * (-oz * dz + h * dz));
	float c = ox * ox + oy * oy - (k * k) * (h * h - 2 * h * oz + oz * oz);
	float disc = b * b - 4 * a * c;
	if (disc < 0)
		return 0;
	float sq = __builtin_sqrtf(disc);
	float t0 = (-b - sq) / (2 * a);
	float t1 = (-b + sq) / (2 * a);
	float z0 = oz + t0 * dz;
	float z1 = oz + t1 * dz;
	if ((z0 > h || z0 < 0) && (z1 > h || z1 < 0))
		return 0;
	else if (z0 > h || z0 < 0)
		t0 = t1;
	ts[0] = t0;
	nc[0] = NC_FROM_T;
	return 1;
}

// Model-space normal of an analytic primitive for the winning candidate.
DEV d3 primitive_normal(CSurface &s, d3 o, d3 d, float t, int ncode)
{
	switch (ncode) {
	case NC_POS_Z: return { 0, 0, 1 };
	case NC_NEG_Z: return { 0, 0, -1 };
	case NC_PLANE_POS: return ld3(s.pn);
	case NC_PLANE_NEG: return vscale(-1, ld3(s.pn));
	default: break;
	}
	switch (s.shape_type) {
	case RTAMD_SHAPE_SPHERE: {
		d3 p = vadd(o, vscale((double)t, d));                 // ray.c:139-148
		return ncode == NC_SPHERE_UNIT ? vnorm(p) : p;
	}
	case RTAMD_SHAPE_CYLINDER: {                               // ray.c:174-180
		double r = (double)s.radius;
		return { (o.x + (double)t * d.x) / r, (o.y + (double)t * d.y) / r, 0 };
	}
	case RTAMD_SHAPE_CONE: {                                   // ray.c:311-315
		float dx = (float)d.x, dy = (float)d.y;
		float ox = (float)o.x, oy = (float)o.y;
		float R = s.radius, h = s.height;
		float sl = __builtin_sqrtf(h * h + R * R);
		float px = ox + t * dx, py = oy + t * dy;
		float rho = __builtin_sqrtf(px * px + py * py);
		float nx = h / sl * px / rho;
		float ny = h / sl * py / rho;
		float nz = R / sl;
		return { (double)nx, (double)ny, (double)nz };
	}
	default:
		return { 0, 0, 0 };
	}
}

// ---- triangles ---------------------------------------------------------------

// ray.c:327-358.  Vertices are float-valued (objreader narrows every coordinate
// to float, mesh.c:93-95), so they are stored as floats and widened here.
DEV bool triangle_intersect(d3 o, d3 d, const float4 &p0, const float4 &p1, const float4 &p2, float wlo, float whi,
		float &aa, float &bb, float &cc, double &t)
{
	d3 u = { (double)p0.x, (double)p0.y, (double)p0.z };
	d3 v = { (double)p1.x, (double)p1.y, (double)p1.z };
	d3 w = { (double)p2.x, (double)p2.y, (double)p2.z };
	d3 edge1 = vsub(v, u);
	d3 edge2 = vsub(w, u);
	d3 pvec = vcross(d, edge2);
	float det = (float)vdot(edge1, pvec);
	d3 tvec = vsub(o, u);
	const double x1 = vdot(tvec, pvec);
	// Cheap rejection before the exact quotients: an approximate b (relative error
	// < 1e-5 whenever it is finite) that is clearly negative or clearly above 1
	// decides the reference's test `bb < 0.0 || bb > 1.0` the same way; anything
	// near the limits, every NaN and every overflow falls through to the exact
	// arithmetic below.
	const float big = 3.0e38f;
	const float rdet = __builtin_amdgcn_rcpf(det);
	const float b_approx = (float)x1 * rdet;
	if ((b_approx < -1e-30f && b_approx > -big) || (b_approx > 1.0001f && b_approx < big))
		return false;
	d3 qvec = vcross(tvec, edge1);
	const double x2 = vdot(d, qvec);
	const double x3 = vdot(edge2, qvec);
	const float c_approx = (float)x2 * rdet;
	const float s_approx = c_approx + b_approx;
	// the caller's window for t, widened (see window_lo / window_hi)
	const float t_approx = (float)x3 * rdet;
	if ((c_approx < -1e-30f && c_approx > -big) ||
			(s_approx > 1.0002f && s_approx < big && __builtin_fabsf(b_approx) < big) ||
			(t_approx < wlo && t_approx > -big) || (t_approx > whi && t_approx < big))
		return false;
	bb = (float)(x1 / (double)det);
	if ((double)bb < 0.0 || (double)bb > 1.0)
		return false;
	cc = (float)(x2 / (double)det);
	if ((double)cc < 0.0 || (double)(cc + bb) > 1.0)
		return false;
	aa = (float)(1.0 - (double)bb - (double)cc);
	t = x3 / (double)det;
	return true;
}

DEV float as_f(int v) { return __builtin_bit_cast(float, v); }
DEV int as_i(float v) { return __builtin_bit_cast(int, v); }
// smallest float strictly above a finite x
DEV float next_up(float x)
{
	if (x == 0.0f)
		return 1.401298464e-45f;
	int i = as_i(x);
	i += x > 0.0f ? 1 : -1;
	return as_f(i);
}

// ---- packet traversal ---------------------------------------------------------
//
// The lanes of a wave that reach a mesh together (64 samples of one pixel, or
// their shadow rays towards the same light) walk the BVH as ONE packet: the node
// and triangle being examined are wave-uniform, so they are fetched through the
// scalar data cache once per wave (constant address space -> s_load) instead of
// once per lane through the vector L1, which is the unit this kernel saturates
// when every lane fetches its own 64-byte node (measured: texture-data unit 96 %
// busy).  The wave keeps one stack of child references in LDS; each lane keeps
// one bit per stack level saying whether its own ray has to visit that entry.
// Results are per lane and identical to the per-lane traversal: every triangle
// whose leaf box a lane's ray may touch inside its window is tested by that
// lane with the same arithmetic.

typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v *const_f4;

DEV const_f4 as_const(const void *p)
{
	return (const_f4)(unsigned long long)p;
}

DEV float4 cload(const_f4 p, int i)
{
	const f4v v = p[i];
	return make_float4(v.x, v.y, v.z, v.w);
}

// Float ray for the packet box tests.  A slab distance (plane - o) / d is
// evaluated as fma(plane, i, k) with i = 1/d (hardware reciprocal, 1 ulp) and
// k = -o * i, both floats.  Against the exact distance T the result is off by at
// most |T| * 2^-21 + |o * i| * 2^-23 (rounding of d, of 1/d, of k and of the fma), so
// for distances inside a window [tlo, thi] the error is below
//   E = max(|tlo|, |thi|) * 2^-20 + max|k| * 2^-22,
// and a box whose true overlap with the window is not empty always satisfies
//   max(entry, tlo) <= min(exit, thi) + 2E     (`slack` = 2E).
struct PacketRay {
	float ix, iy, iz;
	float kx, ky, kz;
	float slack;
};

DEV PacketRay make_packet_ray(d3 o, d3 d, float tlo, float thi)
{
	PacketRay r;
	float fx = (float)d.x, fy = (float)d.y, fz = (float)d.z;
	// a zero component would turn plane * inf + k into NaN
	if (__builtin_fabsf(fx) < 1e-30f) fx = __builtin_copysignf(1e-30f, fx);
	if (__builtin_fabsf(fy) < 1e-30f) fy = __builtin_copysignf(1e-30f, fy);
	if (__builtin_fabsf(fz) < 1e-30f) fz = __builtin_copysignf(1e-30f, fz);
	// 1-ulp reciprocals: i = (1/d)(1 + 2^-22) at worst, inside the error budget above
	r.ix = __builtin_amdgcn_rcpf(fx); r.iy = __builtin_amdgcn_rcpf(fy); r.iz = __builtin_amdgcn_rcpf(fz);
	r.kx = (float)(-(o.x * (double)r.ix));
	r.ky = (float)(-(o.y * (double)r.iy));
	r.kz = (float)(-(o.z * (double)r.iz));
	const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.kx), __builtin_fabsf(r.ky)), __builtin_fabsf(r.kz));
	const float w = __builtin_fmaxf(__builtin_fabsf(tlo), __builtin_fabsf(thi));
	r.slack = 2.0f * (w * 9.5367431640625e-7f + m * 2.384185791015625e-7f);
	return r;
}

// Conservative overlap test of the ray with a box inside [tlo, thi]; tn is the
// (unwidened) entry distance, used only to order the children.
DEV bool packet_box_hit(const PacketRay &r, float lox, float loy, float loz, float hix, float hiy, float hiz,
		float tlo, float thi, float &tn)
{
	float ax = __builtin_fmaf(lox, r.ix, r.kx), bx = __builtin_fmaf(hix, r.ix, r.kx);
	float ay = __builtin_fmaf(loy, r.iy, r.ky), by = __builtin_fmaf(hiy, r.iy, r.ky);
	float az = __builtin_fmaf(loz, r.iz, r.kz), bz = __builtin_fmaf(hiz, r.iz, r.kz);
	tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
			__builtin_fminf(az, bz)), tlo);
	float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
			__builtin_fmaxf(az, bz)), thi);
	return tn <= tf + r.slack;
}

typedef unsigned long long u64;

// The triangle test of ray.c:327-358 for a packet: the cheap rejections are
// evaluated by every lane and decided wave-wide, so the common case (no lane
// can hit) costs no divergent branch; the exact quotients are computed only by
// lanes that may hit.  `act` says whether this lane's ray takes part.
// [wlo, whi] is the caller's acceptance window for t, widened by 2^-18 (window_lo /
// window_hi): a triangle whose approximate t lies outside cannot be accepted, in
// particular the triangle a shadow ray starts on (t = -1e-4 cos), which would
// otherwise reach the exact quotients on every ray.
DEV float window_lo(float tlo) { return tlo - __builtin_fabsf(tlo) * 3.814697265625e-06f - 1e-30f; }
DEV float window_hi(float thi) { return thi + __builtin_fabsf(thi) * 3.814697265625e-06f + 1e-30f; }

DEV bool triangle_intersect_packet(d3 o, d3 d, d3 u, d3 edge1, d3 edge2, bool act, float wlo, float whi,
		float &aa, float &bb, float &cc, double &t)
{
	d3 pvec = vcross(d, edge2);
	float det = (float)vdot(edge1, pvec);
	d3 tvec = vsub(o, u);
	const double x1 = vdot(tvec, pvec);
	const float big = 3.0e38f;
	const float rdet = __builtin_amdgcn_rcpf(det);
	const float b_approx = (float)x1 * rdet;
	bool maybe = act && !((b_approx < -1e-30f && b_approx > -big) || (b_approx > 1.0001f && b_approx < big));
	if (__builtin_amdgcn_ballot_w64(maybe) == 0)
		return false;
	RT_EV(EV_TRI_STAGE2);
	d3 qvec = vcross(tvec, edge1);
	const double x2 = vdot(d, qvec);
	const double x3 = vdot(edge2, qvec);
	const float c_approx = (float)x2 * rdet;
	const float s_approx = c_approx + b_approx;
	// relative error of t_approx below 2^-22 whenever it is finite: far inside the window's margin
	const float t_approx = (float)x3 * rdet;
	maybe = maybe && !((c_approx < -1e-30f && c_approx > -big) ||
			(s_approx > 1.0002f && s_approx < big && __builtin_fabsf(b_approx) < big) ||
			(t_approx < wlo && t_approx > -big) || (t_approx > whi && t_approx < big));
	if (__builtin_amdgcn_ballot_w64(maybe) == 0)
		return false;
	RT_EV(EV_TRI_EXACT);
	bool ok = false;
	if (maybe) {
		bb = (float)(x1 / (double)det);
		if (!((double)bb < 0.0 || (double)bb > 1.0)) {
			cc = (float)(x2 / (double)det);
			if (!((double)cc < 0.0 || (double)(cc + bb) > 1.0)) {
				aa = (float)(1.0 - (double)bb - (double)cc);
				t = x3 / (double)det;
				ok = true;
			}
		}
	}
	return ok;
}

// Nearest triangle of a mesh inside [tnear, tfar] (effective result of
// ray.c:360-462: smallest t among triangles that pass the barycentric tests;
// equal t -> the triangle with the highest index in the mesh's own array, which
// is what the kd leaf loop's `<=` at ray.c:376 gives inside a leaf).
// ANYHIT: stop at the first acceptable triangle (shadow rays only need to know
// whether ray_intersect would return true, shading.c:48).

template <bool ANYHIT>
DEV bool mesh_traverse(const DevScene &sc, CMesh &mesh, d3 o, d3 d, float tnear, float tfar,
		int *stack, int &best_tri, int &best_orig, double &best_t, float &ba, float &bb_, float &bc)
{
	const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes + mesh.node_first);
	const float4 *tris = reinterpret_cast<const float4 *>(sc.tris + mesh.tri_first);
	const PacketRay br = make_packet_ray(o, d, tnear, tfar);
	const double dnear = (double)tnear, dfar = (double)tfar;
	const float wlo = window_lo(tnear);
	bool found = false;
	best_t = __builtin_huge_val();
	best_orig = -1;
	best_tri = -1;
	float limit = tfar;          // float upper bound of the search window, >= best_t
	int sp = 0;
	int ref = mesh.root;
	for (;;) {
		if (ref >= 0) {
			RT_STAT(atomicAdd(&g_stat_nodes, 1ull));
			const float4 q0 = nodes[4 * ref + 0], q1 = nodes[4 * ref + 1];
			const float4 q2 = nodes[4 * ref + 2], q3 = nodes[4 * ref + 3];
			float tn0, tn1;
			bool h0 = packet_box_hit(br, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tnear, limit, tn0);
			bool h1 = packet_box_hit(br, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tnear, limit, tn1);
			int c0 = as_i(q3.x), c1 = as_i(q3.y);
			if (h0 && h1) {
				bool swap = tn1 < tn0;
				int nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
				if (sp < RT_MAX_STACK) {
					stack[sp * RT_BLOCK] = farc;
					sp++;
				}
				ref = nearc;
				continue;
			} else if (h0) {
				ref = c0;
				continue;
			} else if (h1) {
				ref = c1;
				continue;
			}
		} else {
			int v = ~ref;
			int first = v >> 2, count = (v & 3) + 1;
			for (int i = 0; i < count; i++) {
				const float4 p0 = tris[3 * (first + i) + 0];
				const float4 p1 = tris[3 * (first + i) + 1];
				const float4 p2 = tris[3 * (first + i) + 2];
				float aa, bb, cc;
				double t;
				RT_STAT(atomicAdd(&g_stat_tris, 1ull));
				if (!triangle_intersect(o, d, p0, p1, p2, wlo, window_hi(limit), aa, bb, cc, t))
					continue;
				int orig = as_i(p0.w);
				if (t >= dnear && t <= dfar && (t < best_t || (t == best_t && orig > best_orig))) {
					best_t = t;
					best_orig = orig;
					best_tri = first + i;
					ba = aa; bb_ = bb; bc = cc;
					found = true;
					if (ANYHIT)
						return true;
					// next float above or equal to t: boxes further than this cannot win
					float ft = (float)t;
					if ((double)ft < t)
						ft = next_up(ft);
					limit = ft < tfar ? ft : tfar;
				}
			}
		}
		if (sp == 0)
			break;
		sp--;
		ref = stack[sp * RT_BLOCK];
	}
	return found;
}

template <bool ANYHIT>
DEV bool mesh_traverse_packet(const DevScene &sc, CMesh &mesh, d3 o, d3 d, float tnear, float tfar,
		int *wstack_i, int &best_tri, int &best_orig, double &best_t, float &ba, float &bb_, float &bc)
{
	TM2_BEGIN(t_setup);
	int4 *wstack = reinterpret_cast<int4 *>(wstack_i);
	const const_f4 nodes = as_const(sc.nodes + mesh.node_first);
	const CONSTAS DevTriEdges *edges = in_const(sc.edges) + mesh.tri_first;
	const PacketRay pr = make_packet_ray(o, d, tnear, tfar);
	const double dnear = (double)tnear, dfar = (double)tfar;
	const float wlo = window_lo(tnear);
	bool found = false;
	best_t = __builtin_huge_val();
	best_orig = -1;
	best_tri = -1;
	float limit = tfar;
	// wave-uniform state: the lanes whose ray must visit `ref`, the lanes that
	// are finished (any-hit only), the stack depth
	u64 actm = __builtin_amdgcn_ballot_w64(true);
	u64 donem = 0;
	int sp = 0;
	int ref = __builtin_amdgcn_readfirstlane(mesh.root);
	TM2_END(TM_IN_MESH_SETUP, t_setup);
	for (;;) {
		bool at_leaf = true;
		TM2_BEGIN(t_nodes);
		while (ref >= 0) {
			RT_STAT(if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_stat_nodes, 1ull));
			const float4 q0 = cload(nodes, 4 * ref + 0), q1 = cload(nodes, 4 * ref + 1);
			const float4 q2 = cload(nodes, 4 * ref + 2), q3 = cload(nodes, 4 * ref + 3);
			float tn0, tn1;
			const bool h0 = packet_box_hit(pr, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tnear, limit, tn0);
			const bool h1 = packet_box_hit(pr, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tnear, limit, tn1);
			const u64 m0 = __builtin_amdgcn_ballot_w64(h0) & actm;
			const u64 m1 = __builtin_amdgcn_ballot_w64(h1) & actm;
			const int c0 = as_i(q3.x), c1 = as_i(q3.y);
			bool swap = false;
			if (!ANYHIT) {
				// front-to-back for the majority of the lanes
				const u64 pref1 = __builtin_amdgcn_ballot_w64(tn1 < tn0);
				const u64 both = m0 & m1;
				const u64 w1 = (both & pref1) | (m1 & ~m0), w0 = (both & ~pref1) | (m0 & ~m1);
				swap = __builtin_popcountll(w1) > __builtin_popcountll(w0);
			}
			u64 nm = swap ? m1 : m0, fm = swap ? m0 : m1;
			int nc = swap ? c1 : c0, fc = swap ? c0 : c1;
			if (nm == 0) {
				nm = fm;
				nc = fc;
				fm = 0;
			}
			if (nm == 0) {
				at_leaf = false;
				break;
			}
			if (fm != 0) {
				wstack[sp] = make_int4(fc, (int)(unsigned)fm, (int)(unsigned)(fm >> 32), 0);
				sp++;
			}
			ref = nc;
			actm = nm;
		}
		TM2_END(TM_IN_NODES, t_nodes);
		TM2_BEGIN(t_leaves);
		if (at_leaf) {
			const int v = ~ref;
			const int first = v >> 2, count = (v & 3) + 1;
			for (int i = 0; i < count; i++) {
				const CONSTAS DevTriEdges &te = edges[first + i];
				float aa, bb, cc;
				double t;
				RT_STAT(if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_stat_tris, 1ull));
				const bool act = __builtin_amdgcn_inverse_ballot_w64(actm);
				const bool ok = triangle_intersect_packet(o, d, ld3(te.u), ld3(te.e1), ld3(te.e2), act,
						wlo, window_hi(limit), aa, bb, cc, t);
				int orig = te.orig;
				if (ok && t >= dnear && t <= dfar && (t < best_t || (t == best_t && orig > best_orig))) {
					best_t = t;
					best_orig = orig;
					best_tri = first + i;
					ba = aa; bb_ = bb; bc = cc;
					found = true;
					if (!ANYHIT) {
						// next float above or equal to t: boxes further than this cannot win
						float ft = (float)t;
						if ((double)ft < t)
							ft = next_up(ft);
						limit = ft < tfar ? ft : tfar;
					}
				}
				if (ANYHIT) {
					donem = __builtin_amdgcn_ballot_w64(found);
					actm &= ~donem;
					if (actm == 0)
						break;
				}
			}
		}
		TM2_END(TM_IN_LEAVES, t_leaves);
		// next stack entry that some lane still has to visit
		TM2_BEGIN(t_pop);
		for (;;) {
			if (sp == 0) {
				TM2_END(TM_IN_POP, t_pop);
				return found;
			}
			sp--;
			const int4 e = wstack[sp];
			ref = __builtin_amdgcn_readfirstlane(e.x);
			const u64 m = (u64)(unsigned)__builtin_amdgcn_readfirstlane(e.y) |
					((u64)(unsigned)__builtin_amdgcn_readfirstlane(e.z) << 32);
			actm = m & ~donem;
			if (actm != 0)
				break;
		}
		TM2_END(TM_IN_POP, t_pop);
	}
}

// ---- light shafts -------------------------------------------------------------
//
// The n x n shadow rays a lane sends to an area light (shading.c:22-47) all start
// within 1e-4 of the lane's hit point, and the hit points of a wave lie within a
// pixel's footprint.  Every one of these 64 x n^2 segments stays inside the
// "shaft" swept by a box that moves from the bounding box of the wave's origins
// to the bounding box of the light's parallelogram while its half-size changes
// linearly (convexity: a point at fraction s of a segment lies within
// (1-s) r0 + s r1 of the centre line, per axis).  Once per wave and light the
// shaft is tested against the scene: surfaces whose box it misses are skipped by
// all of these rays, and for a mesh the BVH leaves it touches are collected into
// a short list in LDS.  A shadow ray then tests exactly the triangles of the
// listed leaves (after a per-ray box test of each leaf) instead of descending
// the BVH from the root n^2 times.  The outcome per ray is unchanged: a triangle
// the ray can hit lies in a leaf the shaft touches, and the triangle test and
// the acceptance window are the same.

struct ShaftEntry {
	float lox, loy, loz, hix, hiy, hiz;
	int ref, pad;
};

// Per axis, the shaft's upper edge u0 + s gU must reach the box's low plane and
// its lower edge d0 + s gD must not pass the high plane, for one s in [0, 1].
// Each plane gives a bound on s, evaluated as fma(plane, 1/g, -x0/g) like the ray
// slab distances, with the rounding error e = 2^-20 + |x0/g| 2^-22 (window [0,1])
// folded into the constant.  Whether a plane bounds s from below or from above
// depends on the sign of g, which is fixed per shaft: each plane gets one
// coefficient pair for each role, and the pair of the role it does not play is
// (0, -inf) or (0, +inf), which leaves the bound untouched.
struct ShaftAxis {
	float loU_i, loU_k, hiU_i, hiU_k;    // low plane (upper edge): as a lower / as an upper bound
	float loD_i, loD_k, hiD_i, hiD_k;    // high plane (lower edge)
};
struct ShaftTest {
	ShaftAxis x, y, z;
};

struct ShaftState {
	u64 cull;            // bit si: no ray of this wave and light can pass the surface's box
	unsigned counts;     // 8 bits per list: number of leaves, 0xff = no list
	unsigned surfaces;   // 8 bits per list: the surface it belongs to, 0xff = none
	const ShaftEntry *lists;
};

struct f3 { float x, y, z; };

// All shaft arithmetic is float: the half-sizes are inflated by 2^-18 of the
// coordinates' magnitude first, which is far more than the few 2^-24 roundings
// of the operations below can move an edge, and also covers the float tolerance
// of the reference's own box and triangle tests (a ray that misses a box by
// less than that can still be accepted there).
DEV ShaftAxis shaft_axis(float c0, float r0, float c1, float r1)
{
	const float infl = 3.814697265625e-06f;
	r0 += infl * (__builtin_fabsf(c0) + r0);
	r1 += infl * (__builtin_fabsf(c1) + r1);
	ShaftAxis a;
	const float inf = __builtin_huge_valf();
	const float u0 = c0 + r0, d0 = c0 - r0;
	float gU = (c1 + r1) - u0, gD = (c1 - r1) - d0;
	if (__builtin_fabsf(gU) < 1e-30f) gU = __builtin_copysignf(1e-30f, gU);
	if (__builtin_fabsf(gD) < 1e-30f) gD = __builtin_copysignf(1e-30f, gD);
	const float iU = __builtin_amdgcn_rcpf(gU), iD = __builtin_amdgcn_rcpf(gD);
	const float kU = -(u0 * iU), kD = -(d0 * iD);
	const float eU = 9.5367431640625e-7f + __builtin_fabsf(kU) * 4.76837158203125e-7f;
	const float eD = 9.5367431640625e-7f + __builtin_fabsf(kD) * 4.76837158203125e-7f;
	// s gU >= lo - u0: a lower bound on s when gU > 0, an upper bound otherwise
	const bool pU = gU > 0;
	a.loU_i = pU ? iU : 0.0f;  a.loU_k = pU ? kU - eU : -inf;
	a.hiU_i = pU ? 0.0f : iU;  a.hiU_k = pU ? inf : kU + eU;
	// s gD <= hi - d0: an upper bound when gD > 0, a lower bound otherwise
	const bool pD = gD > 0;
	a.hiD_i = pD ? iD : 0.0f;  a.hiD_k = pD ? kD + eD : inf;
	a.loD_i = pD ? 0.0f : iD;  a.loD_k = pD ? -inf : kD - eD;
	return a;
}

DEV ShaftTest shaft_make(f3 c0, f3 r0, f3 c1, f3 r1)
{
	ShaftTest st;
	st.x = shaft_axis(c0.x, r0.x, c1.x, r1.x);
	st.y = shaft_axis(c0.y, r0.y, c1.y, r1.y);
	st.z = shaft_axis(c0.z, r0.z, c1.z, r1.z);
	return st;
}

DEV void shaft_axis_bounds(const ShaftAxis &a, float lo, float hi, float &smin, float &smax)
{
	smin = __builtin_fmaxf(smin, __builtin_fmaxf(__builtin_fmaf(lo, a.loU_i, a.loU_k), __builtin_fmaf(hi, a.loD_i, a.loD_k)));
	smax = __builtin_fminf(smax, __builtin_fminf(__builtin_fmaf(lo, a.hiU_i, a.hiU_k), __builtin_fmaf(hi, a.hiD_i, a.hiD_k)));
}

DEV bool shaft_box(const ShaftTest &st, float lox, float loy, float loz, float hix, float hiy, float hiz)
{
	float smin = 0.0f, smax = 1.0f;
	shaft_axis_bounds(st.x, lox, hix, smin, smax);
	shaft_axis_bounds(st.y, loy, hiy, smin, smax);
	shaft_axis_bounds(st.z, loz, hiz, smin, smax);
	return smin <= smax;
}

// Shaft ends in the model space of an affine surface: the image of a box under
// p -> A p + T lies in the box with centre A c + T and half-size |A| r.
DEV void shaft_to_model(CSurface &s, f3 c, f3 r, f3 &cm, f3 &rm)
{
	const CONSTAS double *m = s.w2m;
	const float m00 = (float)M4(m, 0, 0), m01 = (float)M4(m, 0, 1), m02 = (float)M4(m, 0, 2), m03 = (float)M4(m, 0, 3);
	const float m10 = (float)M4(m, 1, 0), m11 = (float)M4(m, 1, 1), m12 = (float)M4(m, 1, 2), m13 = (float)M4(m, 1, 3);
	const float m20 = (float)M4(m, 2, 0), m21 = (float)M4(m, 2, 1), m22 = (float)M4(m, 2, 2), m23 = (float)M4(m, 2, 3);
	cm.x = m00 * c.x + m01 * c.y + m02 * c.z + m03;
	cm.y = m10 * c.x + m11 * c.y + m12 * c.z + m13;
	cm.z = m20 * c.x + m21 * c.y + m22 * c.z + m23;
	// the matrix entries and the sums are rounded: 2^-20 of the terms' magnitude on top
	const float k = 9.5367431640625e-7f;
	const float bx = r.x + k * __builtin_fabsf(c.x), by = r.y + k * __builtin_fabsf(c.y), bz = r.z + k * __builtin_fabsf(c.z);
	rm.x = __builtin_fabsf(m00) * bx + __builtin_fabsf(m01) * by + __builtin_fabsf(m02) * bz + k * __builtin_fabsf(m03);
	rm.y = __builtin_fabsf(m10) * bx + __builtin_fabsf(m11) * by + __builtin_fabsf(m12) * bz + k * __builtin_fabsf(m13);
	rm.z = __builtin_fabsf(m20) * bx + __builtin_fabsf(m21) * by + __builtin_fabsf(m22) * bz + k * __builtin_fabsf(m23);
}

// Collects the leaves of a mesh whose box the shaft touches.  Everything here
// is wave-uniform.  Returns the number of leaves, or -1 when there are more than
// RT_SHAFT_CAP (or the mesh is a single leaf): the caller then walks the BVH.
DEV int shaft_collect(const DevScene &sc, CMesh &mesh, const ShaftTest &st, int *wstack_i, ShaftEntry *out)
{
	int4 *wstack = reinterpret_cast<int4 *>(wstack_i);
	const const_f4 nodes = as_const(sc.nodes + mesh.node_first);
	const bool writer = lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
	int ref = __builtin_amdgcn_readfirstlane(mesh.root);
	if (ref < 0)
		return -1;
	int sp = 0, n = 0;
	for (;;) {
		RT_STAT(if (writer) atomicAdd(&g_stat_shaft[RT_SHAFT_CAP + 4], 1ull));
		const float4 q0 = cload(nodes, 4 * ref + 0), q1 = cload(nodes, 4 * ref + 1);
		const float4 q2 = cload(nodes, 4 * ref + 2), q3 = cload(nodes, 4 * ref + 3);
		const bool h0 = __builtin_amdgcn_ballot_w64(shaft_box(st, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y)) != 0;
		const bool h1 = __builtin_amdgcn_ballot_w64(shaft_box(st, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w)) != 0;
		const int c0 = as_i(q3.x), c1 = as_i(q3.y);
		int next = 0;
		bool have_next = false;
		if (h0) {
			if (c0 < 0) {
				if (n == RT_SHAFT_CAP)
					return -1;
				if (writer)
					out[n] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, c0, 0 };
				n++;
			} else {
				next = c0;
				have_next = true;
			}
		}
		if (h1) {
			if (c1 < 0) {
				if (n == RT_SHAFT_CAP)
					return -1;
				if (writer)
					out[n] = { q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, c1, 0 };
				n++;
			} else if (have_next) {
				if (sp == RT_MAX_STACK)
					return -1;
				if (writer)
					wstack[sp] = make_int4(c1, 0, 0, 0);
				sp++;
			} else {
				next = c1;
				have_next = true;
			}
		}
		if (!have_next) {
			if (sp == 0)
				return n;
			sp--;
			next = __builtin_amdgcn_readfirstlane(wstack[sp].x);
		}
		ref = next;
	}
}

// Any-hit over a shaft's leaf list: the per-ray part of mesh_traverse_packet<true>
// with the descent replaced by a walk over the listed leaves.
DEV bool mesh_candidates_anyhit(const DevScene &sc, CMesh &mesh, d3 o, d3 d, float tnear, float tfar,
		const ShaftEntry *list, int count)
{
	const CONSTAS DevTriEdges *edges = in_const(sc.edges) + mesh.tri_first;
	const PacketRay pr = make_packet_ray(o, d, tnear, tfar);
	const double dnear = (double)tnear, dfar = (double)tfar;
	const float wlo = window_lo(tnear), whi = window_hi(tfar);
	bool found = false;
	for (int c = 0; c < count; c++) {
		const ShaftEntry e = list[c];
		RT_EV(EV_LIST_BOX);
		float tn;
		const bool h = packet_box_hit(pr, e.lox, e.loy, e.loz, e.hix, e.hiy, e.hiz, tnear, tfar, tn) && !found;
		u64 actm = __builtin_amdgcn_ballot_w64(h);
		if (actm == 0)
			continue;
		const int v = ~__builtin_amdgcn_readfirstlane(e.ref);
		const int first = v >> 2, ntri = (v & 3) + 1;
		for (int i = 0; i < ntri; i++) {
			const CONSTAS DevTriEdges &te = edges[first + i];
			float aa, bb, cc;
			double t;
			RT_STAT(if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_stat_tris, 1ull));
			const bool act = __builtin_amdgcn_inverse_ballot_w64(actm);
			const bool ok = triangle_intersect_packet(o, d, ld3(te.u), ld3(te.e1), ld3(te.e2), act, wlo, whi, aa, bb, cc, t);
			if (ok && t >= dnear && t <= dfar)
				found = true;
			actm &= ~__builtin_amdgcn_ballot_w64(found);
			if (actm == 0)
				break;
		}
		if (__builtin_amdgcn_ballot_w64(!found) == 0)
			break;
	}
	return found;
}

// Builds the shaft of one area light for this wave: `olo`/`ohi` bound the hit
// points of the participating lanes (wave-uniform).  shading.c:22-47 defines the
// segments: origin = hit point + 1e-4 dir, end = origin + (float)|to_light| dir,
// the light point = position + alpha edge1 + beta edge2 with alpha, beta in [0, 2 - 1/n].
DEV ShaftState shaft_build(const DevScene &sc, CLight &light, int n, const float olo[3], const float ohi[3],
		int *wstack, ShaftEntry *lists)
{
	ShaftState ss;
	ss.cull = 0;
	ss.counts = 0xffffffffu;
	ss.surfaces = 0xffffffffu;
	ss.lists = lists;
	// alpha = p/n + drand() reaches (n-1)/n + 1 (shading.c:30-31)
	const float amax = ((float)(n - 1) / (float)n + 1.0f) * 1.000001f;
	const f3 lp = { (float)light.position[0], (float)light.position[1], (float)light.position[2] };
	const f3 e1 = { amax * (float)light.edge1[0], amax * (float)light.edge1[1], amax * (float)light.edge1[2] };
	const f3 e2 = { amax * (float)light.edge2[0], amax * (float)light.edge2[1], amax * (float)light.edge2[2] };
	const f3 llo = { lp.x + __builtin_fminf(0.0f, e1.x) + __builtin_fminf(0.0f, e2.x),
		lp.y + __builtin_fminf(0.0f, e1.y) + __builtin_fminf(0.0f, e2.y),
		lp.z + __builtin_fminf(0.0f, e1.z) + __builtin_fminf(0.0f, e2.z) };
	const f3 lhi = { lp.x + __builtin_fmaxf(0.0f, e1.x) + __builtin_fmaxf(0.0f, e2.x),
		lp.y + __builtin_fmaxf(0.0f, e1.y) + __builtin_fmaxf(0.0f, e2.y),
		lp.z + __builtin_fmaxf(0.0f, e1.z) + __builtin_fmaxf(0.0f, e2.z) };
	const f3 c0 = { 0.5f * (olo[0] + ohi[0]), 0.5f * (olo[1] + ohi[1]), 0.5f * (olo[2] + ohi[2]) };
	const f3 c1 = { 0.5f * (llo.x + lhi.x), 0.5f * (llo.y + lhi.y), 0.5f * (llo.z + lhi.z) };
	// the 1e-4 steps at both ends, the rounding of the segment length to float, and
	// the roundings of the bounds above (2^-20 of the magnitudes involved)
	const float len = __builtin_fabsf(c1.x - c0.x) + __builtin_fabsf(c1.y - c0.y) + __builtin_fabsf(c1.z - c0.z);
	const float k = 9.5367431640625e-7f;
	const float pad0 = 1.1e-4f, pad1 = 1.1e-4f + k * (len + 4.0f);
	const f3 r0 = { 0.5f * (ohi[0] - olo[0]) + pad0 + k * __builtin_fabsf(c0.x), 0.5f * (ohi[1] - olo[1]) + pad0 + k * __builtin_fabsf(c0.y),
		0.5f * (ohi[2] - olo[2]) + pad0 + k * __builtin_fabsf(c0.z) };
	const f3 r1 = { 0.5f * (lhi.x - llo.x) + pad1 + k * (__builtin_fabsf(llo.x) + __builtin_fabsf(lhi.x)),
		0.5f * (lhi.y - llo.y) + pad1 + k * (__builtin_fabsf(llo.y) + __builtin_fabsf(lhi.y)),
		0.5f * (lhi.z - llo.z) + pad1 + k * (__builtin_fabsf(llo.z) + __builtin_fabsf(lhi.z)) };
	const ShaftTest world = shaft_make(c0, r0, c1, r1);
	int slot = 0;
	for (int si = 0; si < sc.num_surfaces; si++) {
		CSurface &s = in_const(sc.surfaces)[si];
		const bool is_mesh = s.shape_type == RTAMD_SHAPE_MESH;
		const int my_slot = slot;
		if (is_mesh)
			slot++;
		const bool touches = __builtin_amdgcn_ballot_w64(shaft_box(world, s.bbox[0], s.bbox[1], s.bbox[2],
				s.bbox[3], s.bbox[4], s.bbox[5])) != 0;
		RT_STAT(const bool counter = lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)));
		if (!touches) {
			if (si < 64)
				ss.cull |= 1ull << si;
			RT_STAT(if (counter) atomicAdd(&g_stat_shaft[RT_SHAFT_CAP + 2], 1ull));
			continue;
		}
		if (!is_mesh || my_slot >= RT_SHAFT_LISTS || !s.affine || si >= 255) {
			RT_STAT(if (counter) atomicAdd(&g_stat_shaft[RT_SHAFT_CAP + 3], 1ull));
			continue;
		}
		f3 cm0, rm0, cm1, rm1;
		shaft_to_model(s, c0, r0, cm0, rm0);
		shaft_to_model(s, c1, r1, cm1, rm1);
		const ShaftTest model = shaft_make(cm0, rm0, cm1, rm1);
		const int n_leaves = shaft_collect(sc, in_const(sc.meshes)[s.mesh], model, wstack, lists + my_slot * RT_SHAFT_CAP);
		RT_STAT(if (counter) atomicAdd(&g_stat_shaft[n_leaves >= 0 ? n_leaves : RT_SHAFT_CAP + 1], 1ull));
		if (n_leaves >= 0) {
			ss.counts = (ss.counts & ~(0xffu << (8 * my_slot))) | ((unsigned)n_leaves << (8 * my_slot));
			ss.surfaces = (ss.surfaces & ~(0xffu << (8 * my_slot))) | ((unsigned)si << (8 * my_slot));
		}
	}
	return ss;
}

// ---- ray.c:483-578 + 643-672 ---------------------------------------------------

DEV ShaftState shaft_none()
{
	ShaftState ss;
	ss.cull = 0;
	ss.counts = 0xffffffffu;
	ss.surfaces = 0xffffffffu;
	ss.lists = nullptr;
	return ss;
}

// `shaft` is passed by value: taking its address would pin it in scratch memory.
template <bool ANYHIT, bool PACKET>
DEV bool trace(const DevScene &sc, const Ray &ray, int *stack, Best &best, const ShaftState shaft)
{
	best.surface = -1;
	best.tri = -1;
	best.t = __builtin_huge_valf();
	// surfaces the light's shaft misses are stepped over without being looked at
	const bool masked = ANYHIT && PACKET;
	const u64 wanted = ~shaft.cull;
	auto next_wanted = [&](int i) {
		if (masked && i < 64) {
			const u64 rest = wanted >> i;
			return rest ? i + (int)__builtin_ctzll(rest) : 64;
		}
		return i;
	};
	for (int si = next_wanted(0); si < sc.num_surfaces; si = next_wanted(si + 1)) {
		CSurface &s = in_const(sc.surfaces)[si];
		int shaft_count = 0xff, shaft_slot = 0;
		if (masked) {
			for (int k = 0; k < RT_SHAFT_LISTS; k++)
				if ((int)((shaft.surfaces >> (8 * k)) & 0xffu) == si) {
					shaft_count = (int)((shaft.counts >> (8 * k)) & 0xffu);
					shaft_slot = k;
				}
		}
		RT_EV(ANYHIT ? EV_BBOX_SHADOW : EV_BBOX_PRIMARY);
		float bnear, bfar;
		TM2_BEGIN(t_bbox);
		const bool in_box = surface_bbox_test(s.bbox, ray, bnear, bfar);
		TM2_END(TM_IN_BBOX, t_bbox);
		if (!in_box)
			continue;
		TM2_BEGIN(t_xform);
		d3 o = xform_point(s, ray.o);
		d3 d = xform_dir(s, ray.d);
		TM2_END(TM_IN_XFORM, t_xform);
		TM2_BEGIN(t_prim);
		float ts[2] = { -__builtin_huge_valf(), -__builtin_huge_valf() };
		int nc[2] = { NC_FROM_T, NC_FROM_T };
		int hits;
		int tri = -1;
		float ba = 0, bb = 0, bc = 0;
		if (s.shape_type != RTAMD_SHAPE_MESH)
			RT_EV(ANYHIT ? EV_PRIM_SHADOW : EV_PRIM_PRIMARY);
		switch (s.shape_type) {
		case RTAMD_SHAPE_PLANE: hits = plane_intersect(s, o, d, ts, nc); break;
		case RTAMD_SHAPE_DISK: hits = disk_intersect(s, o, d, bnear, ts, nc); break;
		case RTAMD_SHAPE_SPHERE: hits = sphere_intersect(s, o, d, ts, nc); break;
		case RTAMD_SHAPE_CYLINDER: hits = cylinder_intersect(s, o, d, ts, nc); break;
		case RTAMD_SHAPE_CONE: hits = cone_intersect(s, o, d, ts, nc); break;
		default: {
			int orig;
			double t;
			if (ANYHIT && PACKET && shaft_count != 0xff) {
				RT_EV(EV_MESH_LIST_SHADOW);
				// every listed leaf's triangles; a hit's t lies in [bnear, bfar]
				hits = mesh_candidates_anyhit(sc, in_const(sc.meshes)[s.mesh], o, d, bnear, bfar,
						shaft.lists + shaft_slot * RT_SHAFT_CAP, shaft_count) ? 1 : 0;
				t = (double)bnear;
			} else if (PACKET) {
				RT_EV(ANYHIT ? EV_MESH_WALK_SHADOW : EV_MESH_WALK_PRIMARY);
				hits = mesh_traverse_packet<ANYHIT>(sc, in_const(sc.meshes)[s.mesh], o, d, bnear, bfar, stack,
						tri, orig, t, ba, bb, bc) ? 1 : 0;
			} else
				hits = mesh_traverse<ANYHIT>(sc, in_const(sc.meshes)[s.mesh], o, d, bnear, bfar, stack,
						tri, orig, t, ba, bb, bc) ? 1 : 0;
			ts[0] = (float)t;                       // ray.c:471
			nc[0] = NC_MESH;
			break;
		}
		}
		TM2_END(TM_IN_PRIM, t_prim);
		if (hits == 0)
			continue;
		TM2_BEGIN(t_select);
		float t;
		int ncode;
		if (hits == 1) {                             // ray.c:530-536
			if (ts[0] < bnear || ts[0] > bfar)
				continue;
			t = ts[0];
			ncode = nc[0];
		} else {                                     // ray.c:537-567
			bool t0_ok = ts[0] >= bnear && ts[0] <= bfar;
			bool t1_ok = ts[1] >= bnear && ts[1] <= bfar;
			if (!t0_ok && !t1_ok)
				continue;
			else if (t0_ok && !t1_ok) { t = ts[0]; ncode = nc[0]; }
			else if (!t0_ok && t1_ok) { t = ts[1]; ncode = nc[1]; }
			else if (ts[0] < ts[1]) { t = ts[0]; ncode = nc[0]; }
			else { t = ts[1]; ncode = nc[1]; }
		}
		if (ANYHIT) {
			best.surface = si;
			return true;
		}
		// ray.c:661: first hit or strictly smaller t
		if (best.surface < 0 || (double)t < (double)best.t) {
			best.surface = si;
			best.t = t;
			best.ncode = ncode;
			best.tri = tri;
			best.a = ba; best.b = bb; best.c = bc;
		}
		TM2_END(TM_IN_SELECT, t_select);
	}
	return best.surface >= 0;
}

// Position and world-space normal of the winning hit, ray.c:574-576.
DEV void finish_hit(const DevScene &sc, const Ray &ray, const Best &best, d3 &position, d3 &normal)
{
	CSurface &s = in_const(sc.surfaces)[best.surface];
	position = vadd(ray.o, vscale((double)best.t, ray.d));
	d3 n;
	if (best.ncode == NC_MESH) {
		// ray.c:472-475
		CMesh &mesh = in_const(sc.meshes)[s.mesh];
		const float4 *nrm = reinterpret_cast<const float4 *>(sc.normals + mesh.tri_first) + 3 * best.tri;
		const float4 a = nrm[0], b = nrm[1], c = nrm[2];
		d3 n0 = { (double)a.x, (double)a.y, (double)a.z };
		d3 n1 = { (double)b.x, (double)b.y, (double)b.z };
		d3 n2 = { (double)c.x, (double)c.y, (double)c.z };
		n = vadd(vadd(vscale((double)best.a, n0), vscale((double)best.b, n1)), vscale((double)best.c, n2));
	} else {
		d3 o = xform_point(s, ray.o);
		d3 d = xform_dir(s, ray.d);
		n = primitive_normal(s, o, d, best.t, best.ncode);
	}
	normal = vnorm(xform_normal(s.w2m, n));
}

// ---- texture.c:72-116, 135-182 -------------------------------------------------

DEV c3 texel_at(const DevScene &sc, const DevCubeFace &f, int x, int y)
{
	const float4 v = reinterpret_cast<const float4 *>(sc.texels)[f.offset + (int64_t)f.width * y + x];
	return { v.x, v.y, v.z };
}

DEV c3 texture_texel(const DevScene &sc, int face, double u, double v)
{
	const DevCubeFace f = sc.faces[face];
	if (u == 1.0) u -= 1e-6;
	if (v == 1.0) v -= 1e-6;
	u = u - __builtin_floor(u);
	v = v - __builtin_floor(v);
	u *= f.width;
	v *= f.height;
	int x0 = (int)__builtin_floor(u);
	int y0 = (int)__builtin_floor(v);
	double ualpha = u - x0, valpha = v - y0;
	double ubeta = 1 - ualpha, vbeta = 1 - valpha;
	int x1 = (x0 == f.width - 1) ? x0 : x0 + 1;
	int y1 = (y0 == f.height - 1) ? y0 : y0 + 1;
	// out-of-range coordinates can only come from a NaN direction; keep the fetch in bounds
	x0 = min(max(x0, 0), f.width - 1);  x1 = min(max(x1, 0), f.width - 1);
	y0 = min(max(y0, 0), f.height - 1); y1 = min(max(y1, 0), f.height - 1);
	c3 caa = texel_at(sc, f, x0, y0), cab = texel_at(sc, f, x1, y0);
	c3 cba = texel_at(sc, f, x0, y1), cbb = texel_at(sc, f, x1, y1);
	return cadd(cscale((float)vbeta, cadd(cscale((float)ubeta, caa), cscale((float)ualpha, cab))),
			cscale((float)valpha, cadd(cscale((float)ubeta, cba), cscale((float)ualpha, cbb))));
}

DEV c3 cubemap_colour(const DevScene &sc, d3 d)
{
	int face = -1;
	float u = 0.0f, v = 0.0f;
	double dx = __builtin_fabs(d.x), dy = __builtin_fabs(d.y), dz = __builtin_fabs(d.z);
	if (dx >= dy && dx >= dz) {
		if (d.x >= 0) { u = (float)((1.0 + d.z / d.x) / 2.0); v = (float)((1.0 + d.y / d.x) / 2.0); face = 1; }
		else { u = (float)(1.0 - (1.0 - d.z / d.x) / 2.0); v = (float)((1.0 - d.y / d.x) / 2.0); face = 0; }
	} else if (dy >= dz && dy >= dx) {
		if (d.y >= 0) { u = (float)((1.0 + d.x / d.y) / 2.0); v = (float)(1.0 - (1.0 - d.z / d.y) / 2.0); face = 3; }
		else { u = (float)((1.0 - d.x / d.y) / 2.0); v = (float)((1.0 + d.z / d.y) / 2.0); face = 2; }
	} else if (dz >= dx && dz >= dy) {
		if (d.z >= 0) { u = (float)(1.0 - (1.0 + d.x / d.z) / 2.0); v = (float)((1.0 + d.y / d.z) / 2.0); face = 5; }
		else { u = (float)((1.0 - d.x / d.z) / 2.0); v = (float)((1.0 - d.y / d.z) / 2.0); face = 4; }
	}
	if (face < 0)
		return { 0.0f, 0.0f, 0.0f };
	return texture_texel(sc, face, (double)u, (double)v);
}

// ---- lighting.c ----------------------------------------------------------------

// lighting.c:7-19
DEV c3 diff_colour(CLight &light, CMaterial &mat, d3 light_dir, d3 normal)
{
	double dp = vdot(light_dir, normal);
	float ndotv = (float)(0 > dp ? 0.0 : dp);
	c3 light_col = cscale(light.intensity, cscale(ndotv, lc3(light.colour)));
	return cmul(lc3(mat.diffuse), light_col);
}

// lighting.c:21-34 (called with cam_dir and light_dir swapped, shading.c:52; the
// half vector is symmetric in them)
DEV c3 spec_colour(CLight &light, CMaterial &mat, d3 first, d3 second, d3 normal)
{
	d3 half = vnorm(vadd(first, second));
	double dp = vdot(half, normal);
	float base = (float)(0 > dp ? 0.0 : dp);
	float hdotn = __builtin_powf(base, (float)mat.shininess);
	c3 light_col = cscale(light.intensity, cscale(hdotn, lc3(light.colour)));
	return cmul(lc3(mat.specular), light_col);
}

// ---- wave helpers ------------------------------------------------------------


// Adds per-lane counts into a global counter with one atomic per wave.
DEV void wave_count_add(unsigned long long *counter, unsigned value)
{
	unsigned total = value;
	for (int off = 32; off > 0; off >>= 1)
		total += __shfl_xor(total, off, 64);
	if (lane_id() == 0 && total)
		atomicAdd(counter, (unsigned long long)total);
}

DEV long long to_fixed(float c)
{
	// NaN and infinities cannot be accumulated; the quantiser maps them to 0 / 255 anyway
	if (!(c == c)) return 0;
	double v = (double)c;
	if (v > 1.0e6) v = 1.0e6;
	if (v < -1.0e6) v = -1.0e6;
	return __double2ll_rn(v * RT_FIX_SCALE);
}

DEV void accumulate(long long *accum, uint32_t slot, c3 c)
{
	unsigned long long *p = reinterpret_cast<unsigned long long *>(accum + 3ull * slot);
	long long r = to_fixed(c.r), g = to_fixed(c.g), b = to_fixed(c.b);
	if (r) atomicAdd(p + 0, (unsigned long long)r);
	if (g) atomicAdd(p + 1, (unsigned long long)g);
	if (b) atomicAdd(p + 2, (unsigned long long)b);
}

// What a path vertex needs besides its ray.
struct ShadeIO {
	const DevScene *sc;
	int rng_mode;
	QueueItem *queue_out;
	unsigned *queue_count;
	unsigned queue_capacity;
	DevCounters *counters;
	ShaftEntry *shaft_lists;     // this wave's leaf lists in LDS, or null
	int *wave_stack;             // transposed light loops: this wave's packet stack in LDS
	float *wave_sum;             // ... and 384 floats for the ordered sums
};

// ---- light samples -------------------------------------------------------------

// shading.c:24-37: the point of the light that sample j aims at.
DEV d3 light_sample_point(const DevScene &sc, CLight &light, bool area, int n, int li, uint32_t key, int j)
{
	const d3 lpos = ld3(light.position);
	if (!area)
		return lpos;
	int p = j % n, q = j / n;
	float alpha = (float)((double)((float)p / (float)n) +
			keyed_drand(key, PURPOSE_LIGHT0 + (uint32_t)li, 2u * (uint32_t)j));
	float beta = (float)((double)((float)q / (float)n) +
			keyed_drand(key, PURPOSE_LIGHT0 + (uint32_t)li, 2u * (uint32_t)j + 1u));
	return vadd(vadd(lpos, vscale((double)alpha, ld3(light.edge1))), vscale((double)beta, ld3(light.edge2)));
}

// shading.c:39-46
DEV Ray shadow_ray_towards(d3 position, d3 light_pos, d3 &light_dir)
{
	d3 to_light = vsub(light_pos, position);
	light_dir = vnorm(to_light);
	Ray shadow;
	shadow.d = light_dir;
	shadow.o = vadd(position, vscale(1e-4, light_dir));
	shadow.tnear = 0;
	shadow.tfar = (float)vlen(to_light);
	return shadow;
}

DEV double readlane_f64(double v, int lane)
{
	const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
	return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
DEV d3 readlane_d3(d3 v, int lane) { return { readlane_f64(v.x, lane), readlane_f64(v.y, lane), readlane_f64(v.z, lane) }; }

// The light loops of a wave whose lanes hold UNRELATED hit points (bounced rays).
// Per lane, the shadow rays of one hit walk the BVH one after the other and the
// lanes of the wave diverge.  Transposed, the wave takes one hit at a time and
// its lanes are that hit's light samples: up to 64 shadow rays that leave the
// same point towards the same light, i.e. a coherent packet (scalar node and
// triangle fetches, one traversal for all of them).  The samples' contributions
// are then added in sample order (shading.c:18-54 adds them one by one).  Which
// form is cheaper depends on samples per light (m) against hits in the wave (H):
// transposed costs H packet traversals per 64 samples, per lane costs m
// divergent ones; the form is chosen per wave and light.
#ifndef RT_TRANSPOSE_GAIN
#define RT_TRANSPOSE_GAIN 4     // a divergent per-lane step is taken to cost this many packet steps
#endif

DEV d3 shfl_d3(d3 v, int src)
{
	return { __shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64) };
}

#ifndef RT_TRANSPOSE_MAX_GROUP
#define RT_TRANSPOSE_MAX_GROUP 8     // most hit points whose samples share one packet
#endif

template <typename Unused = void>
DEV c3 lights_transposed(const ShadeIO &io, bool lit, d3 position, d3 normal, d3 cam_dir, int material,
		uint32_t key, int *lane_stack, unsigned &n_shadow)
{
	const DevScene &sc = *io.sc;
	const unsigned lane = lane_id();
	float *sum = io.wave_sum;
	c3 total = { 0.0f, 0.0f, 0.0f };
	const u64 lit_lanes = __builtin_amdgcn_ballot_w64(lit);
	const int hits = __builtin_popcountll(lit_lanes);
	for (int li = 0; li < sc.num_lights; li++) {
		CLight &light = in_const(sc.lights)[li];
		const bool area = light.type == RTAMD_LIGHT_AREA;
		const int n = area ? sc.shadow_samples : 1;
		const int m = n * n;
		const float inv = (float)(1.0 / (double)(n * n));
		// hit points per packet: as many runs of m lanes as fit into the wave
		const int group = m >= 64 ? 1 : 64 / m;
		if (group > RT_TRANSPOSE_MAX_GROUP || (m >= 64 && m * RT_TRANSPOSE_GAIN < hits)) {
			// per lane, as shade_vertex does it
			if (lit) {
				CMaterial &mat = in_const(sc.materials)[material];
				c3 light_total = { 0.0f, 0.0f, 0.0f };
				for (int j = 0; j < m; j++) {
					d3 light_dir;
					const Ray shadow = shadow_ray_towards(position, light_sample_point(sc, light, area, n, li, key, j), light_dir);
					n_shadow++;
					Best dummy;
					if (trace<true, false>(sc, shadow, lane_stack, dummy, shaft_none()))
						continue;
					light_total = cadd(light_total, cadd(diff_colour(light, mat, light_dir, normal),
							spec_colour(light, mat, cam_dir, light_dir, normal)));
				}
				total = cadd(total, cscale(inv, light_total));
			}
			continue;
		}
		if (m < 64) {
			// `group` hit points at a time: lane = (run, sample)
			const int run = (int)lane / m, j = (int)lane - run * m;
			u64 todo = lit_lanes;
			while (todo) {
				const u64 members = todo;
				int src = 0, taken = 0;
				for (int k = 0; k < group && todo; k++) {
					const int hk = __builtin_ctzll(todo);
					todo &= todo - 1;
					if (run == k)
						src = hk;
					taken++;
				}
				const bool valid = run < taken;
				// the hit point of this lane's run
				const d3 hp = shfl_d3(position, src), hn = shfl_d3(normal, src), hc = shfl_d3(cam_dir, src);
				const uint32_t hk = (uint32_t)__shfl((int)key, src, 64);
				const int hm = __shfl(material, src, 64);
				c3 c = { 0.0f, 0.0f, 0.0f };
				if (valid) {
					CMaterial &mat = in_const(sc.materials)[hm];
					d3 light_dir;
					const Ray shadow = shadow_ray_towards(hp, light_sample_point(sc, light, area, n, li, hk, j), light_dir);
					Best dummy;
					if (!trace<true, true>(sc, shadow, io.wave_stack, dummy, shaft_none()))
						c = cadd(diff_colour(light, mat, light_dir, hn), spec_colour(light, mat, hc, light_dir, hn));
				}
				// per run and channel: ((0 + c_0) + c_1) + ... in sample order (shading.c:18-54;
				// an occluded sample adds +0, which leaves the non-negative sum unchanged)
				sum[lane] = c.r; sum[64 + lane] = c.g; sum[128 + lane] = c.b;
				for (int task = (int)lane; task < 3 * taken; task += 64) {
					const int r = task / 3, ch = task - 3 * r;
					const float *src_run = sum + 64 * ch + r * m;
					float a = src_run[0];
					for (int i = 1; i < m; i++)
						a += src_run[i];
					sum[192 + task] = a;
				}
				u64 rest = members;
				for (int k = 0; k < taken; k++) {
					const int hk2 = __builtin_ctzll(rest);
					rest &= rest - 1;
					if ((int)lane == hk2) {
						total = cadd(total, cscale(inv, { sum[192 + 3 * k], sum[193 + 3 * k], sum[194 + 3 * k] }));
						n_shadow += (unsigned)m;
					}
				}
			}
			continue;
		}
		// 64 or more samples per light: one hit point at a time, 64 samples per packet
		u64 todo = lit_lanes;
		while (todo) {
			const int h = __builtin_ctzll(todo);
			todo &= todo - 1;
			const d3 hp = readlane_d3(position, h), hn = readlane_d3(normal, h), hc = readlane_d3(cam_dir, h);
			const uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)key, h);
			CMaterial &mat = in_const(sc.materials)[__builtin_amdgcn_readlane(material, h)];
			c3 acc = { 0.0f, 0.0f, 0.0f };      // running sum, identical in every lane
			for (int base = 0; base < m; base += 64) {
				const int j = base + (int)lane;
				c3 c = { 0.0f, 0.0f, 0.0f };
				if (j < m) {
					d3 light_dir;
					const Ray shadow = shadow_ray_towards(hp, light_sample_point(sc, light, area, n, li, hk, j), light_dir);
					Best dummy;
					if (!trace<true, true>(sc, shadow, io.wave_stack, dummy, shaft_none()))
						c = cadd(diff_colour(light, mat, light_dir, hn), spec_colour(light, mat, hc, light_dir, hn));
				}
				const int cnt = m - base < 64 ? m - base : 64;
				sum[lane] = c.r; sum[64 + lane] = c.g; sum[128 + lane] = c.b;
				if (lane < 3) {
					float a = lane == 0 ? acc.r : lane == 1 ? acc.g : acc.b;
					for (int i = 0; i < cnt; i++)
						a += sum[64 * lane + i];
					sum[192 + lane] = a;
				}
				acc = { sum[192], sum[193], sum[194] };
			}
			if ((int)lane == h) {
				total = cadd(total, cscale(inv, acc));
				n_shadow += (unsigned)m;
			}
		}
	}
	return total;
}

// ---- shading of one path vertex: shading.c:114-141 without the recursion ------
//
// Returns the radiance this vertex contributes directly (environment, or the sum
// of the light loops) and pushes reflection / glossy children into the bounce
// queue with their throughput weight.


// `ray_again` returns the incoming direction once more (recomputed or re-read by
// the caller): it is needed again only when a reflection is spawned, after all the
// shadow rays, and is not worth six registers until then.
template <bool PACKET, bool TRANSPOSE, typename RayAgain>
DEV c3 shade_vertex(const ShadeIO &io, const Ray &ray, uint32_t depth, uint32_t key, c3 weight,
		uint32_t slot, bool active, int *stack, rtamd_hit_record *rec,
		unsigned &n_shadow, unsigned &n_env, RayAgain ray_again)
{
	const DevScene &sc = *io.sc;
	c3 total = { 0.0f, 0.0f, 0.0f };
	Best best;
	bool hit = false;
	d3 position = { 0, 0, 0 }, normal = { 0, 0, 0 };
	if (active) {
		TM_BEGIN(t_trace);
		hit = trace<false, PACKET>(sc, ray, stack, best, shaft_none());
		TM_END(TM_TRACE_PRIMARY, t_trace);
		TM_BEGIN(t_finish);
		if (hit)
			finish_hit(sc, ray, best, position, normal);
		TM_END(TM_FINISH, t_finish);
	}
	// bounds of the wave's hit points (origins of its shadow rays), rounded outwards
	float olo[3] = { 0, 0, 0 }, ohi[3] = { 0, 0, 0 };
	const bool shafts = RT_SHAFT && PACKET && io.shaft_lists && sc.any_area_light;
	TM_BEGIN(t_bounds);
	if (shafts) {
		const float inf = __builtin_huge_valf();
		const double pv[3] = { position.x, position.y, position.z };
		for (int a = 0; a < 3; a++) {
			const float f = (float)pv[a];
			const float w = __builtin_fabsf(f) * 1.1920929e-7f + 1e-37f;
			float lo = hit ? f - w : inf, hi = hit ? f + w : -inf;
			for (int off = 32; off > 0; off >>= 1) {
				lo = __builtin_fminf(lo, __shfl_xor(lo, off, 64));
				hi = __builtin_fmaxf(hi, __shfl_xor(hi, off, 64));
			}
			olo[a] = as_f(__builtin_amdgcn_readfirstlane(as_i(lo)));
			ohi[a] = as_f(__builtin_amdgcn_readfirstlane(as_i(hi)));
		}
	}
	TM_END(TM_BOUNDS, t_bounds);
	if (rec && active) {
		rtamd_hit_record r;
		r.surface = hit ? best.surface : -1;
		r.triangle = -1;
		if (hit && best.ncode == NC_MESH) {
			CMesh &mesh = in_const(sc.meshes)[in_const(sc.surfaces)[best.surface].mesh];
			r.triangle = as_i(reinterpret_cast<const float4 *>(sc.tris + mesh.tri_first)[3 * best.tri].w);
		}
		r.t = hit ? best.t : 0.0f;
		r.pad = 0.0f;
		r.position[0] = position.x; r.position[1] = position.y; r.position[2] = position.z;
		r.normal[0] = normal.x; r.normal[1] = normal.y; r.normal[2] = normal.z;
		*rec = r;
	}

	int n_children = 0;
	CMaterial *mat = nullptr;
	d3 cam_dir = { 0, 0, 0 };
	int material = 0;
	if (active && !hit) {
		// shading.c:123-129
		if (sc.has_cubemap) {
			TM_BEGIN(t_env);
			total = cubemap_colour(sc, ray.d);
			TM_END(TM_ENV, t_env);
			n_env++;
		} else
			total = lc3(sc.background);
	} else if (active) {
		material = in_const(sc.surfaces)[best.surface].material;
		mat = &in_const(sc.materials)[material];
		cam_dir = vnorm(vscale(-1, ray.d));                          // shading.c:118
		// shading.c:133-135 -> hit_light_colour, shading.c:7-58
		for (int li = 0; !TRANSPOSE && li < sc.num_lights; li++) {
			CLight &light = in_const(sc.lights)[li];
			const bool area = light.type == RTAMD_LIGHT_AREA;
			const int n = area ? sc.shadow_samples : 1;
			c3 light_total = { 0.0f, 0.0f, 0.0f };
			ShaftState shaft_state = shaft_none();
			const bool use_shaft = shafts && area && n >= 2;
			if (use_shaft) {
				TM_BEGIN(t_shaft);
				shaft_state = shaft_build(sc, light, n, olo, ohi, stack, io.shaft_lists);
				TM_END(TM_SHAFT, t_shaft);
			}
			for (int j = 0; j < n * n; j++) {
				TM_BEGIN(t_lsetup);
				d3 light_dir;
				const Ray shadow = shadow_ray_towards(position, light_sample_point(sc, light, area, n, li, key, j), light_dir);
				n_shadow++;
				RT_EV(EV_SHADOW_RAYS);
				Best dummy;
				TM_END(TM_LIGHT_SETUP, t_lsetup);
				TM_BEGIN(t_shadow);
#ifdef RT_EXPERIMENT_NO_SIMT_SHADOWS
				const bool occluded = PACKET ? trace<true, PACKET>(sc, shadow, stack, dummy, shaft_state) : false;
#else
				const bool occluded = trace<true, PACKET>(sc, shadow, stack, dummy, shaft_state);
#endif
				TM_END(TM_TRACE_SHADOW, t_shadow);
				if (occluded)
					continue;
				TM_BEGIN(t_shade);
				RT_EV(EV_SHADE);
				c3 diff_col = diff_colour(light, *mat, light_dir, normal);
				c3 spec_col = spec_colour(light, *mat, cam_dir, light_dir, normal);
				light_total = cadd(light_total, cadd(diff_col, spec_col));
				TM_END(TM_SHADE, t_shade);
			}
			light_total = cscale((float)(1.0 / (double)(n * n)), light_total);
			total = cadd(total, light_total);
		}
		// shading.c:70-112: children exist when the material reflects and the child's
		// depth passes the cut at shading.c:120
		if (mat->reflect > 0.0f && (int)depth + 1 <= sc.max_reflections)
			n_children = (mat->glossiness <= 0.0f || depth > 1) ? 1 : sc.reflection_samples;
	}
	if (TRANSPOSE) {
		// every lane of the wave takes part, whether it holds a hit or not
		const c3 lit_total = lights_transposed(io, active && hit, position, normal, cam_dir, material, key, stack, n_shadow);
		if (active && hit)
			total = lit_total;
	}

	// Compact the spawning lanes: one slot reservation per wave.
	TM_BEGIN(t_spawn);
	if (io.queue_out) {
		unsigned want = (unsigned)n_children;
		unsigned prefix = want;
		const unsigned lane = lane_id();
		// inclusive prefix sum over the wave
		for (int off = 1; off < 64; off <<= 1) {
			unsigned v = __shfl_up(prefix, off, 64);
			if (lane >= (unsigned)off)
				prefix += v;
		}
		unsigned wave_total = __shfl(prefix, 63, 64);
		if (wave_total) {
			unsigned base = 0;
			if (lane == 0)
				base = atomicAdd(io.queue_count, wave_total);
			base = __shfl(base, 0, 64);
			unsigned my = base + prefix - want;
			if (want) {
				if (my + want > io.queue_capacity) {
					atomicAdd(&io.counters->queue_overflow, 1ull);
				} else {
					// shading.c:81-84
					const d3 in_dir = ray_again();
					double dn = vdot(in_dir, normal);
					d3 rdir = vadd(in_dir, vscale(-2 * dn, normal));          // vector.c:77-81
					d3 rorg = vadd(position, vscale(1e-2, rdir));
					c3 w = cmul(lc3(mat->specular), { mat->reflect, mat->reflect, mat->reflect });
					w = cmul(w, weight);
					if (want == 1 && (mat->glossiness <= 0.0f || depth > 1)) {
						QueueItem it;
						it.o[0] = rorg.x; it.o[1] = rorg.y; it.o[2] = rorg.z;
						it.d[0] = rdir.x; it.d[1] = rdir.y; it.d[2] = rdir.z;
						it.weight[0] = w.r; it.weight[1] = w.g; it.weight[2] = w.b;
						it.slot = slot; it.key = key_child(key, 0); it.depth = depth + 1;
						it.pad[0] = it.pad[1] = 0;
						io.queue_out[my] = it;
					} else {
						// shading.c:93-109
						const float inv = (float)(1. / (double)sc.reflection_samples);
						w = cscale(inv, w);
						d3 pdir = vnorm(rdir);
						const d3 n1 = { 1, 0, 0 }, n2 = { 0, 1, 0 };
						d3 orth = (__builtin_fabs(vdot(pdir, n1)) < vlen(pdir) / 1.4142135623730950488016887242096980785696718753769480731766)
								? vcross(pdir, n1) : vcross(pdir, n2);            // shading.c:60-68
						d3 ta = vnorm(orth);
						d3 tb = vnorm(vcross(pdir, ta));
						for (unsigned i = 0; i < want; i++) {
							double ra = keyed_drand(key, PURPOSE_GLOSSY, 2u * i);
							double rb = keyed_drand(key, PURPOSE_GLOSSY, 2u * i + 1u);
							d3 a = vscale((double)mat->glossiness * (2 * ra - 1), ta);
							d3 b = vscale((double)mat->glossiness * (2 * rb - 1), tb);
							d3 dir = vadd(pdir, vadd(a, b));
							QueueItem it;
							it.o[0] = rorg.x; it.o[1] = rorg.y; it.o[2] = rorg.z;
							it.d[0] = dir.x; it.d[1] = dir.y; it.d[2] = dir.z;
							it.weight[0] = w.r; it.weight[1] = w.g; it.weight[2] = w.b;
							it.slot = slot; it.key = key_child(key, i); it.depth = depth + 1;
							it.pad[0] = it.pad[1] = 0;
							io.queue_out[my + i] = it;
						}
					}
				}
			}
		}
	}
	TM_END(TM_SPAWN, t_spawn);
	return total;
}

// Sample index -> pixel.  Samples are numbered tile by tile over the tiles this
// call owns, pixels row-major inside a tile, `spp` consecutive samples per pixel.
struct TileMap {
	int tile_w, tile_h, tiles_x, tiles_y;
	int tile_first, tile_stride, owned;
	int x0, y0, x1, y1;
};

DEV bool sample_to_pixel(const TileMap &tm, int spp, unsigned long long sample, int &px, int &py, int &k,
		uint32_t &slot)
{
	unsigned long long pix = sample / (unsigned)spp;
	k = (int)(sample - pix * (unsigned)spp);
	unsigned per_tile = (unsigned)(tm.tile_w * tm.tile_h);
	unsigned long long lt = pix / per_tile;
	unsigned in_tile = (unsigned)(pix - lt * per_tile);
	slot = (uint32_t)pix;
	if (lt >= (unsigned long long)tm.owned)
		return false;
	int tile = tm.tile_first + (int)lt * tm.tile_stride;
	int tx = tile % tm.tiles_x, ty = tile / tm.tiles_x;
	px = tx * tm.tile_w + (int)(in_tile % (unsigned)tm.tile_w);
	py = ty * tm.tile_h + (int)(in_tile / (unsigned)tm.tile_w);
	return px >= tm.x0 && px < tm.x1 && py >= tm.y0 && py < tm.y1;
}

// ray.c:19-44 with the jitter of ray.c:47-59 or the pixel centre of ray.c:62-65
DEV Ray camera_ray(const DevScene &sc, int i, int j, float offx, float offy)
{
	const int nx = sc.width, ny = sc.height;
	float u = (float)(sc.cam_left + sc.cam_width * (double)((float)i + offx) / (double)nx);
	float v = (float)(sc.cam_bottom + sc.cam_height * (double)((float)j + offy) / (double)ny);
	float d = sc.cam_d;
	Ray r;
	r.o = ld3(sc.cam_pos);
	r.d = vnorm(vadd(vscale((double)(-d), ld3(sc.cam_w)),
				vadd(vscale((double)u, ld3(sc.cam_u)), vscale((double)v, ld3(sc.cam_v)))));
	r.tnear = 0;
	r.tfar = __builtin_huge_valf();
	return r;
}

DEV void sample_offsets(const DevScene &sc, int rng_mode, const int32_t *glibc_table, uint32_t seed,
		int px, int py, int k, uint32_t &key, float &offx, float &offy)
{
	uint32_t pixel = (uint32_t)(sc.width * py + px);
	key = key0_of(seed, pixel, (uint32_t)k);
	if (!sc.antialiasing) {
		offx = 0.5f;
		offy = 0.5f;
		return;
	}
	const int aa = sc.aa_samples;
	const float n = (float)aa;
	int p = k % aa, q = k / aa;
	double r0, r1;
	if (rng_mode == RTAMD_RNG_GLIBC) {
		// draw 2*((j*W+i)*aa^2+k)+{0,1} of the serial stream
		unsigned long long idx = 2ull * ((unsigned long long)pixel * (unsigned)sc.spp + (unsigned)k);
		r0 = (double)glibc_table[idx] / ((double)2147483647);
		r1 = (double)glibc_table[idx + 1] / ((double)2147483647);
	} else {
		r0 = keyed_drand(key, PURPOSE_AA, 0);
		r1 = keyed_drand(key, PURPOSE_AA, 1);
	}
	offx = (float)(((double)p + r0) / (double)n);
	offy = (float)(((double)q + r1) / (double)n);
}

} // namespace

// =============================================================================
// Kernels
// =============================================================================

// Primary samples: camera ray, nearest hit, direct lighting with shadow rays,
// environment lookup; children go to the bounce queue.  raytracer.c:32-54,78-84.
// Register budget: the primary kernel is fastest at five waves per SIMD (<= 96
// VGPRs, some spilling; measured 78.9 ms vs 83.9 ms at four and 81.5 ms at six
// on the 1080p showcase frame); the bounce kernel keeps per-lane stacks in LDS
// (32 KB per workgroup) and runs at four.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_MIN_WAVES_BOUNCE
#define RT_MIN_WAVES_BOUNCE 4
#endif
#ifndef RT_BOUNCE_PACKET
#define RT_BOUNCE_PACKET 0
#endif
#ifndef RT_BOUNCE_TRANSPOSE
#define RT_BOUNCE_TRANSPOSE 1
#endif
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES)
k_primary(DevScene sc, TileMap tm, unsigned long long sample_first, unsigned long long sample_count,
		int rng_mode, uint32_t seed, const int32_t *glibc_table, long long *accum, float *direct_out,
		rtamd_hit_record *hits, QueueItem *queue_out, unsigned *queue_count, unsigned queue_capacity,
		DevCounters *counters)
{
	// one traversal stack per wave (packet traversal)
	__shared__ int4 s_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ __attribute__((aligned(16))) float s_sum[RT_BLOCK / 64][3 * 64];
	const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar register
	int *stack = reinterpret_cast<int *>(s_stack + wave * RT_MAX_STACK);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		s_timing[threadIdx.x >> 6][threadIdx.x & 63] = 0;
#endif
	TM_BEGIN(t_kernel);
	TM_BEGIN(t_setup);
	unsigned long long local = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
	unsigned long long sample = sample_first + local;
	int px = 0, py = 0, k = 0;
	uint32_t slot = 0;
	bool active = local < sample_count && sample_to_pixel(tm, sc.spp, sample, px, py, k, slot);

	uint32_t key = 0;
	Ray ray;
	ray.o = { 0, 0, 0 }; ray.d = { 0, 0, 1 }; ray.tnear = 0; ray.tfar = 0;
	if (active) {
		float offx, offy;
		sample_offsets(sc, rng_mode, glibc_table, seed, px, py, k, key, offx, offy);
		ray = camera_ray(sc, px, py, offx, offy);
	}
	__shared__ ShaftEntry s_shaft[RT_BLOCK / 64][RT_SHAFT_LISTS * RT_SHAFT_CAP];
	ShadeIO io = { &sc, rng_mode, queue_out, queue_count, queue_capacity, counters, s_shaft[wave], nullptr, nullptr };
	rtamd_hit_record *rec = nullptr;
	if (hits && active) {
		// window-local record index
		unsigned long long wpix = (unsigned long long)(py - tm.y0) * (unsigned)(tm.x1 - tm.x0) + (unsigned)(px - tm.x0);
		rec = hits + wpix * (unsigned)sc.spp + (unsigned)k;
	}
	unsigned n_shadow = 0, n_env = 0;
	c3 weight = { 1.0f, 1.0f, 1.0f };
	TM_END(TM_SETUP, t_setup);
	auto ray_again = [&]() {
		// same arithmetic from the same inputs: the same bits as `ray.d`
		unsigned tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		int qx = 0, qy = 0, qk = 0;
		uint32_t qslot, qkey;
		float ox, oy;
		sample_to_pixel(tm, sc.spp, sample_first + (unsigned long long)blockIdx.x * RT_BLOCK + tid, qx, qy, qk, qslot);
		sample_offsets(sc, rng_mode, glibc_table, seed, qx, qy, qk, qkey, ox, oy);
		return camera_ray(sc, qx, qy, ox, oy).d;
	};
	c3 total = shade_vertex<true, false>(io, ray, 0, key, weight, slot, active, stack, rec, n_shadow, n_env, ray_again);
	// the pixel mapping again, rather than five registers kept through the kernel
	int tx = 0, ty = 0, tk = 0;
	uint32_t tslot = 0;
	bool tactive;
	{
		unsigned tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		const unsigned long long l2 = (unsigned long long)blockIdx.x * RT_BLOCK + tid;
		tactive = l2 < sample_count && sample_to_pixel(tm, sc.spp, sample_first + l2, tx, ty, tk, tslot);
	}

	TM_BEGIN(t_tail);
	const unsigned lane = lane_id();
	if (64 % sc.spp == 0) {
		// All samples of a pixel sit in consecutive lanes of this wave: add them in
		// the reference's order, ((c0 + c1) + c2) + ... in float (raytracer.c:41-45).
		const unsigned spp = (unsigned)sc.spp;
		const unsigned base = lane - lane % spp;
		float ar = total.r, ag = total.g, ab = total.b;
		if (spp % 4 == 0) {
			// through LDS: one lane per (pixel, channel) walks its run of samples
			float *sw = s_sum[wave];
			sw[lane] = total.r; sw[64 + lane] = total.g; sw[128 + lane] = total.b;
			const unsigned ppw = 64 / spp;
			if (lane < 3 * ppw) {
				float *run = sw + 64 * (lane / ppw) + (lane % ppw) * spp;
				const float4 *src = reinterpret_cast<const float4 *>(run);
				float4 v = src[0];
				float acc = v.x;
				acc += v.y; acc += v.z; acc += v.w;
				for (unsigned j = 1; j < spp / 4; j++) {
					v = src[j];
					acc += v.x; acc += v.y; acc += v.z; acc += v.w;
				}
				run[0] = acc;
			}
			ar = sw[base]; ag = sw[64 + base]; ab = sw[128 + base];
		} else {
			for (unsigned j = 1; j < spp; j++) {
				ar += __shfl(total.r, (int)(base + j), 64);
				ag += __shfl(total.g, (int)(base + j), 64);
				ab += __shfl(total.b, (int)(base + j), 64);
			}
		}
		if (tactive && lane == base) {
			if (direct_out) {
				// no other contribution can arrive: write the pixel (raytracer.c:46,81)
				if (sc.antialiasing) {
					const float s = (float)(1.0 / (double)sc.spp);
					ar = s * ar; ag = s * ag; ab = s * ab;
				}
				float4 out = { ar, ag, ab, 0.0f };
				reinterpret_cast<float4 *>(direct_out)[(unsigned long long)sc.width * ty + tx] = out;
			} else {
				accumulate(accum, tslot, { ar, ag, ab });
			}
		}
	} else {
		// Per-pixel sum in 2^-32 fixed point: integer adds commute, so the result does
		// not depend on the order in which lanes, waves or later bounces arrive.
		long long r = tactive ? to_fixed(total.r) : 0, g = tactive ? to_fixed(total.g) : 0,
		     b = tactive ? to_fixed(total.b) : 0;
		// segmented reduction over runs of lanes that share a pixel (runs are contiguous)
		uint32_t seg = tactive ? tslot : 0xffffffffu;
		for (int off = 1; off < 64; off <<= 1) {
			uint32_t oseg = __shfl_down(seg, off, 64);
			long long orr = __shfl_down(r, off, 64), og = __shfl_down(g, off, 64), ob = __shfl_down(b, off, 64);
			// lane + off belongs to my run only if every lane in between does, which holds
			// for contiguous runs when its slot equals mine
			if (lane + off < 64 && oseg == seg) {
				r += orr; g += og; b += ob;
			}
		}
		uint32_t prev = __shfl_up(seg, 1, 64);
		bool head = tactive && (lane == 0 || prev != seg);
		if (head) {
			unsigned long long *p = reinterpret_cast<unsigned long long *>(accum + 3ull * tslot);
			if (r) atomicAdd(p + 0, (unsigned long long)r);
			if (g) atomicAdd(p + 1, (unsigned long long)g);
			if (b) atomicAdd(p + 2, (unsigned long long)b);
		}
	}
	const u64 active_lanes = __builtin_amdgcn_ballot_w64(tactive), env_lanes = __builtin_amdgcn_ballot_w64(n_env != 0);
	if (lane == 0) {
		if (active_lanes) atomicAdd(&counters->rays_primary, (unsigned long long)__builtin_popcountll(active_lanes));
		if (env_lanes) atomicAdd(&counters->env_lookups, (unsigned long long)__builtin_popcountll(env_lanes));
	}
	wave_count_add(&counters->rays_shadow, n_shadow);
	TM_END(TM_TAIL, t_tail);
	TM_END(TM_KERNEL, t_kernel);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		atomicAdd(&g_timing[threadIdx.x & 63], s_timing[threadIdx.x >> 6][threadIdx.x & 63]);
#endif
}

// One bounce: every queued ray is traced and shaded like a primary one
// (shading.c:114-141 at depth >= 1) and its radiance, scaled by the path
// weight, is added to its pixel.  Grid-stride over the queue population, which
// is read from device memory so the host never waits between bounces.
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_BOUNCE)
k_bounce(DevScene sc, const QueueItem *queue_in, const unsigned *queue_in_count, unsigned queue_capacity,
		long long *accum, QueueItem *queue_out, unsigned *queue_out_count, DevCounters *counters)
{
#if RT_BOUNCE_PACKET
	// consecutive queue entries come from the lanes of one spawning wave (one
	// pixel's samples): they are coherent enough to walk the BVH as a packet
	__shared__ int4 s_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ ShaftEntry s_shaft[RT_BLOCK / 64][RT_SHAFT_LISTS * RT_SHAFT_CAP];
	int *stack = reinterpret_cast<int *>(s_stack + (threadIdx.x >> 6) * RT_MAX_STACK);
	ShaftEntry *shaft_lists = s_shaft[threadIdx.x >> 6];
#else
	__shared__ int s_stack[RT_MAX_STACK * RT_BLOCK];
	int *stack = s_stack + threadIdx.x;
	ShaftEntry *shaft_lists = nullptr;
#endif
	unsigned count = *queue_in_count;
	if (count > queue_capacity)
		count = queue_capacity;
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicMax(&counters->queue_peak, (unsigned long long)count);
	const unsigned stride = gridDim.x * RT_BLOCK;
	// every lane of a wave runs the same number of iterations (the wave-level
	// compaction inside shade_vertex needs all 64 lanes present)
	const unsigned rounds = (count + stride - 1) / stride;
	__shared__ int4 s_wave_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ float s_wave_sum[RT_BLOCK / 64][384];
	ShadeIO io = { &sc, RTAMD_RNG_KEYED, queue_out, queue_out_count, queue_capacity, counters, shaft_lists,
		reinterpret_cast<int *>(s_wave_stack + (threadIdx.x >> 6) * RT_MAX_STACK), s_wave_sum[threadIdx.x >> 6] };
	unsigned n_shadow = 0, n_env = 0, n_rays = 0;
	for (unsigned round = 0; round < rounds; round++) {
		unsigned idx = round * stride + blockIdx.x * RT_BLOCK + threadIdx.x;
		bool active = idx < count;
		Ray ray;
		ray.o = { 0, 0, 0 }; ray.d = { 0, 0, 1 }; ray.tnear = 0; ray.tfar = 0;
		c3 weight = { 0, 0, 0 };
		uint32_t slot = 0, key = 0, depth = 0;
		if (active) {
			const QueueItem it = queue_in[idx];
			ray.o = { it.o[0], it.o[1], it.o[2] };
			ray.d = { it.d[0], it.d[1], it.d[2] };
			ray.tnear = 0;
			ray.tfar = __builtin_huge_valf();
			weight = { it.weight[0], it.weight[1], it.weight[2] };
			slot = it.slot; key = it.key; depth = it.depth;
			n_rays++;
		}
		auto ray_again = [&]() {
			unsigned i = idx;
			asm volatile("" : "+v"(i));
			const double *dq = queue_in[i].d;
			return d3{ dq[0], dq[1], dq[2] };
		};
		c3 total = shade_vertex<RT_BOUNCE_PACKET != 0, RT_BOUNCE_TRANSPOSE != 0>(io, ray, depth, key, weight, slot, active, stack, nullptr, n_shadow, n_env, ray_again);
		if (active)
			accumulate(accum, slot, cmul(weight, total));
	}
	wave_count_add(&counters->rays_secondary, n_rays);
	wave_count_add(&counters->rays_shadow, n_shadow);
	wave_count_add(&counters->env_lookups, n_env);
}

#if defined(RT_TRAVERSAL_STATS) || defined(RT_TIMING)
__global__ void k_collect_stats(DevCounters *counters)
{
#ifdef RT_TRAVERSAL_STATS
	counters->bvh_nodes = g_stat_nodes;
	counters->tri_tests = g_stat_tris;
	g_stat_nodes = 0;
	g_stat_tris = 0;
	for (int i = 0; i < RT_SHAFT_CAP + 8; i++) {
		if (g_stat_shaft[i])
			printf("shaft %d %llu\n", i, g_stat_shaft[i]);
		g_stat_shaft[i] = 0;
	}
	for (int i = 0; i < 16; i++) {
		if (g_stat_ev[i])
			printf("event %d %llu\n", i, g_stat_ev[i]);
		g_stat_ev[i] = 0;
	}
#endif
#ifdef RT_TIMING
	for (int i = 0; i < TM_COUNT; i++) {
		printf("timing %d %llu\n", i, g_timing[i]);
		g_timing[i] = 0;
	}
#endif
}
#endif

// Accumulators -> the reference's float RGBA frame (raytracer.c:46,81): the
// per-pixel sum is scaled by the float 1.0/aa^2 when antialiasing is on.
__global__ void k_resolve(DevScene sc, TileMap tm, unsigned long long pixel_first, unsigned long long pixel_count,
		const long long *accum, float *colour)
{
	unsigned long long local = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (local >= pixel_count)
		return;
	unsigned long long pix = pixel_first + local;
	int px, py, k;
	uint32_t slot;
	if (!sample_to_pixel(tm, 1, pix, px, py, k, slot))
		return;
	const long long *a = accum + 3ull * slot;
	float r = (float)((double)a[0] / RT_FIX_SCALE);
	float g = (float)((double)a[1] / RT_FIX_SCALE);
	float b = (float)((double)a[2] / RT_FIX_SCALE);
	if (sc.antialiasing) {
		float s = (float)(1.0 / (double)sc.spp);
		r = s * r; g = s * g; b = s * b;
	}
	float4 out = { r, g, b, 0.0f };
	reinterpret_cast<float4 *>(colour)[(unsigned long long)sc.width * py + px] = out;
}

// ray_intersect on a batch of caller-supplied rays (ray.h:25, ray.c:643-672).
__global__ void __launch_bounds__(RT_BLOCK)
k_trace(DevScene sc, const double *rays, unsigned long long n, rtamd_hit_record *hits)
{
	__shared__ int s_stack[RT_MAX_STACK * RT_BLOCK];
	int *stack = s_stack + threadIdx.x;
	unsigned long long i = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
	if (i >= n)
		return;
	const double *p = rays + 8 * i;
	Ray ray;
	ray.o = { p[0], p[1], p[2] };
	ray.d = { p[3], p[4], p[5] };
	ray.tnear = (float)p[6];
	ray.tfar = (float)p[7];
	Best best;
	rtamd_hit_record r;
	r.surface = -1; r.triangle = -1; r.t = 0; r.pad = 0;
	r.position[0] = r.position[1] = r.position[2] = 0;
	r.normal[0] = r.normal[1] = r.normal[2] = 0;
	if (trace<false, false>(sc, ray, stack, best, shaft_none())) {
		d3 position, normal;
		finish_hit(sc, ray, best, position, normal);
		r.surface = best.surface;
		if (best.ncode == NC_MESH) {
			CMesh &mesh = in_const(sc.meshes)[in_const(sc.surfaces)[best.surface].mesh];
			r.triangle = as_i(reinterpret_cast<const float4 *>(sc.tris + mesh.tri_first)[3 * best.tri].w);
		}
		r.t = best.t;
		r.position[0] = position.x; r.position[1] = position.y; r.position[2] = position.z;
		r.normal[0] = normal.x; r.normal[1] = normal.y; r.normal[2] = normal.z;
	}
	hits[i] = r;
}

// camera_ray / camera_ray_aa for every sample of the selection (ray.h:23-24).
__global__ void k_camera_rays(DevScene sc, TileMap tm, unsigned long long sample_count, int rng_mode,
		uint32_t seed, const int32_t *glibc_table, double *rays)
{
	unsigned long long sample = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (sample >= sample_count)
		return;
	int px, py, k;
	uint32_t slot;
	if (!sample_to_pixel(tm, sc.spp, sample, px, py, k, slot))
		return;
	uint32_t key;
	float offx, offy;
	sample_offsets(sc, rng_mode, glibc_table, seed, px, py, k, key, offx, offy);
	Ray ray = camera_ray(sc, px, py, offx, offy);
	double *o = rays + 8 * (((unsigned long long)sc.width * py + px) * (unsigned)sc.spp + (unsigned)k);
	o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z;
	o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z;
	o[6] = (double)ray.tnear; o[7] = (double)ray.tfar;
}

// Frame <-> dense per-owner tile buffer (multi-GPU framebuffer exchange).
__global__ void k_tiles_copy(int width, int height, TileMap tm, float *frame, float *packed, int unpack)
{
	unsigned long long pix = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned per_tile = (unsigned)(tm.tile_w * tm.tile_h);
	if (pix >= (unsigned long long)tm.owned * per_tile)
		return;
	int px, py, k;
	uint32_t slot;
	TileMap whole = tm;
	whole.x0 = 0; whole.y0 = 0; whole.x1 = width; whole.y1 = height;
	if (!sample_to_pixel(whole, 1, pix, px, py, k, slot))
		return;
	float4 *f = reinterpret_cast<float4 *>(frame) + (unsigned long long)width * py + px;
	float4 *p = reinterpret_cast<float4 *>(packed) + pix;
	if (unpack)
		*f = *p;
	else
		*p = *f;
}

// =============================================================================
// Host-side launchers (called from api.hip)
// =============================================================================

static TileMap make_tile_map(int width, int height, const rtamd_render_params &p)
{
	TileMap tm;
	tm.tile_w = p.tile_w;
	tm.tile_h = p.tile_h;
	tm.tiles_x = (width + p.tile_w - 1) / p.tile_w;
	tm.tiles_y = (height + p.tile_h - 1) / p.tile_h;
	tm.tile_first = p.tile_first;
	tm.tile_stride = p.tile_stride;
	int total = tm.tiles_x * tm.tiles_y;
	tm.owned = p.tile_first < total ? (total - p.tile_first + p.tile_stride - 1) / p.tile_stride : 0;
	bool whole = p.x0 == 0 && p.y0 == 0 && p.x1 == 0 && p.y1 == 0;
	tm.x0 = whole ? 0 : p.x0;
	tm.y0 = whole ? 0 : p.y0;
	tm.x1 = whole ? width : p.x1;
	tm.y1 = whole ? height : p.y1;
	return tm;
}

int rt_tiles_owned(int width, int height, const rtamd_render_params &p)
{
	return make_tile_map(width, height, p).owned;
}

unsigned long long rt_owned_pixels(int width, int height, const rtamd_render_params &p)
{
	TileMap tm = make_tile_map(width, height, p);
	return (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h);
}

hipError_t rt_launch_primary(const DevScene &sc, const rtamd_render_params &p,
		unsigned long long sample_first, unsigned long long sample_count, const int32_t *glibc_table,
		long long *accum, float *direct_out, rtamd_hit_record *hits, QueueItem *queue_out,
		unsigned *queue_count, unsigned queue_capacity, DevCounters *counters, hipStream_t stream)
{
	if (!sample_count)
		return hipSuccess;
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long blocks = (sample_count + RT_BLOCK - 1) / RT_BLOCK;
	hipLaunchKernelGGL(k_primary, dim3((unsigned)blocks), dim3(RT_BLOCK), 0, stream, sc, tm, sample_first,
			sample_count, p.rng_mode, p.seed, glibc_table, accum, direct_out, hits, queue_out, queue_count,
			queue_capacity, counters);
	return hipGetLastError();
}

hipError_t rt_launch_bounce(const DevScene &sc, const QueueItem *queue_in, const unsigned *queue_in_count,
		unsigned queue_capacity, long long *accum, QueueItem *queue_out, unsigned *queue_out_count,
		DevCounters *counters, unsigned grid_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL(k_bounce, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, queue_in, queue_in_count,
			queue_capacity, accum, queue_out, queue_out_count, counters);
	return hipGetLastError();
}

hipError_t rt_launch_resolve(const DevScene &sc, const rtamd_render_params &p, unsigned long long pixel_first,
		unsigned long long pixel_count, const long long *accum, float *colour, DevCounters *counters,
		hipStream_t stream)
{
	if (!pixel_count)
		return hipSuccess;
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long blocks = (pixel_count + 255) / 256;
	hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, stream, sc, tm, pixel_first, pixel_count,
			accum, colour);
	return hipGetLastError();
}

// Build options only: moves the traversal statistics into the counter block.
hipError_t rt_launch_collect_stats(DevCounters *counters, hipStream_t stream)
{
#if defined(RT_TRAVERSAL_STATS) || defined(RT_TIMING)
	hipLaunchKernelGGL(k_collect_stats, dim3(1), dim3(1), 0, stream, counters);
	return hipGetLastError();
#else
	(void)counters; (void)stream;
	return hipSuccess;
#endif
}

hipError_t rt_launch_trace(const DevScene &sc, const double *rays, unsigned long long n,
		rtamd_hit_record *hits, hipStream_t stream)
{
	if (!n)
		return hipSuccess;
	unsigned long long blocks = (n + RT_BLOCK - 1) / RT_BLOCK;
	hipLaunchKernelGGL(k_trace, dim3((unsigned)blocks), dim3(RT_BLOCK), 0, stream, sc, rays, n, hits);
	return hipGetLastError();
}

hipError_t rt_launch_camera_rays(const DevScene &sc, const rtamd_render_params &p,
		const int32_t *glibc_table, double *rays, hipStream_t stream)
{
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long samples = (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h) * (unsigned)sc.spp;
	if (!samples)
		return hipSuccess;
	unsigned long long blocks = (samples + 255) / 256;
	hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)blocks), dim3(256), 0, stream, sc, tm, samples, p.rng_mode,
			p.seed, glibc_table, rays);
	return hipGetLastError();
}

hipError_t rt_launch_tiles_copy(int width, int height, const rtamd_render_params &p, float *frame,
		float *packed, int unpack, hipStream_t stream)
{
	TileMap tm = make_tile_map(width, height, p);
	unsigned long long pixels = (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h);
	if (!pixels)
		return hipSuccess;
	unsigned long long blocks = (pixels + 255) / 256;
	hipLaunchKernelGGL(k_tiles_copy, dim3((unsigned)blocks), dim3(256), 0, stream, width, height, tm, frame,
			packed, unpack);
	return hipGetLastError();
}
