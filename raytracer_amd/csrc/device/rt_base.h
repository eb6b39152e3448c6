// Build switches, measurement hooks, the reference's vector / colour / matrix
// arithmetic (vector.c, colour.c, matrix.c), ray and candidate records, the keyed
// sample generator.
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

// Light shafts (see shaft_build): compile-time switches and capacities.
#ifndef RT_SHAFT
#define RT_SHAFT 1
#endif
#ifndef RT_SHAFT_POOL
#define RT_SHAFT_POOL 96       // leaves a wave's lists hold together (at most 254); a shaft that needs more falls back to the BVH walk
#endif
#define RT_SHAFT_CAP RT_SHAFT_POOL
#ifndef RT_SHAFT_LISTS
#define RT_SHAFT_LISTS 4       // mesh surfaces (in traversal order) that get a list
#endif

#ifndef RT_ANYHIT_ORDER
// Per-lane shadow walks (any hit will do): 0 nearer child first, 1 the child the ray spends
// more of its length in, 2 farther child first.  Measured on 8-glossy 640 x 640 x 64 with
// every level walked per lane: 264 / 259 / 250 ms - occluders of a bounced hit's shadow
// rays lie away from the hit point more often than next to it.
#define RT_ANYHIT_ORDER 2
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
__device__ unsigned long long g_stat_shaft[RT_SHAFT_CAP + 16];
// wave-level event counts: see the EV_ names
__device__ unsigned long long g_stat_ev[40];
// list walks: [0] walks, [1] listed leaves, [2] leaves some lane's ray enters, [3] the most leaves one
// lane's ray enters (summed over walks), [4] leaves entered summed over lanes, [5] lanes in the walks
__device__ unsigned long long g_stat_list[8];
// per-lane any-hit walks by length: [2k] unoccluded, [2k + 1] occluded, k = bit length of the step count
__device__ unsigned long long g_stat_walk[32];
enum { EV_BBOX_PRIMARY, EV_BBOX_SHADOW, EV_PRIM_PRIMARY, EV_PRIM_SHADOW, EV_MESH_WALK_PRIMARY, EV_MESH_WALK_SHADOW,
	EV_MESH_LIST_SHADOW, EV_LIST_BOX, EV_TRI_STAGE2, EV_TRI_EXACT, EV_SHADE, EV_SHADOW_RAYS,
	// per-lane walks: walks started (per wave) and the lanes in them, loop iterations and the lanes in them
	EV_LANE_WALKS, EV_LANE_WALK_LANES, EV_LANE_STEPS, EV_LANE_STEP_LANES,
	EV_BOX_SIZE0 = 16 /* .. 31: largest extent of a wave's hit-point box, 0.001 * 2^k */,
	// light samples that add exactly nothing (lanes), and packets of samples all of that kind
	EV_DARK = 32, EV_DARK_PACKETS };
#define RT_EV_ADD(e, n) do { const unsigned long long n_ = (unsigned long long)(n); if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_stat_ev[e], n_); } while (0)
#define RT_EV(e) RT_EV_ADD(e, 1)
#else
#define RT_EV(e) do { } while (0)
#define RT_EV_ADD(e, n) do { } while (0)
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
__device__ unsigned long long g_timing[TM_COUNT], g_timing_bounce[TM_COUNT];
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

