// Light shafts: per wave and area light, which surfaces and which BVH leaves
// its shadow rays can reach.
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

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

// Collects the leaves of a mesh whose box the shaft touches.  Returns the number of
// leaves, or -1 when there are more than `cap` (or the mesh is a single leaf): the
// caller then walks the BVH.
//
// The descent starts depth first with everything wave-uniform (scalar node fetches, one
// node per round): the thin shaft of a pixel on a floor is done after some 15 nodes.  A
// shaft that is not done after RT_SHAFT_SERIAL_VISITS nodes is wide - a wave next to a
// mesh touches hundreds - and the rest of its descent is breadth first and shared out
// over the lanes: the pending nodes become a frontier in LDS, lane i tests the two
// children of frontier node i (every lane holds the same shaft), and ballots with prefix
// counts place the touched leaves in the list and the touched inner nodes in the next
// frontier, so that a round takes a whole level of the tree.  A level of more than 64
// touched nodes counts as an overflow.  The caller's lanes need not all be active: the
// active ones share the work by rank.
#ifndef RT_SHAFT_SERIAL_VISITS
#define RT_SHAFT_SERIAL_VISITS 24
#endif
DEV int shaft_collect(const DevScene &sc, CMesh &mesh, const ShaftTest &st, int *wstack, ShaftEntry *out, int cap)
{
	static_assert(4 * RT_MAX_STACK >= 128 && RT_SHAFT_SERIAL_VISITS < 64, "a wave's stack area holds two frontiers of 64 nodes");
	const u64 exec = __builtin_amdgcn_ballot_w64(true);
	const u64 below = (1ull << lane_id()) - 1ull;
	const int rank = __builtin_popcountll(exec & below), lanes = __builtin_popcountll(exec);
	const bool writer = rank == 0;
	int ref = __builtin_amdgcn_readfirstlane(mesh.root);
	if (ref < 0)
		return -1;
	int sp = 0, n = 0, nf = 0;
	{
		const const_f4 nodes = as_const(sc.nodes + mesh.node_first);
		for (int visits = 0;; visits++) {
			if (visits == RT_SHAFT_SERIAL_VISITS) {
				// the pending nodes, wstack[0 .. sp) and `ref`, are the first frontier
				if (writer)
					wstack[sp] = ref;
				nf = sp + 1;
				break;
			}
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
					if (n == cap)
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
					if (n == cap)
						return -1;
					if (writer)
						out[n] = { q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, c1, 0 };
					n++;
				} else if (have_next) {
					// (the serial part pushes at most once per visit: sp stays below 64)
					if (writer)
						wstack[sp] = c1;
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
				next = __builtin_amdgcn_readfirstlane(wstack[sp]);
			}
			ref = next;
		}
	}
	// two frontiers of 64 node references in the wave's stack area (128 ints)
	const float4 *nodes = reinterpret_cast<const float4 *>(sc.nodes + mesh.node_first);
	int *front = wstack, *next = wstack + 64;
	while (nf > 0) {
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		int ni = 0;
		for (int base = 0; base < nf; base += lanes) {
			const bool active = base + rank < nf;
			bool h0 = false, h1 = false;
			int c0 = 0, c1 = 0;
			float4 q0 = { 0, 0, 0, 0 }, q1 = q0, q2 = q0;
			if (active) {
				const int node = front[base + rank];
				RT_STAT(atomicAdd(&g_stat_shaft[RT_SHAFT_CAP + 5], 1ull));
				q0 = nodes[4 * node + 0]; q1 = nodes[4 * node + 1]; q2 = nodes[4 * node + 2];
				const float4 q3 = nodes[4 * node + 3];
				c0 = as_i(q3.x); c1 = as_i(q3.y);
				h0 = shaft_box(st, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
				h1 = shaft_box(st, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
			}
			const u64 l0 = __builtin_amdgcn_ballot_w64(h0 && c0 < 0), l1 = __builtin_amdgcn_ballot_w64(h1 && c1 < 0);
			const u64 i0 = __builtin_amdgcn_ballot_w64(h0 && c0 >= 0), i1 = __builtin_amdgcn_ballot_w64(h1 && c1 >= 0);
			const int nl0 = __builtin_popcountll(l0), nl = nl0 + __builtin_popcountll(l1);
			const int ni0 = __builtin_popcountll(i0), nin = ni0 + __builtin_popcountll(i1);
			if (n + nl > cap || ni + nin > 64)
				return -1;
			if (h0 && c0 < 0)
				out[n + __builtin_popcountll(l0 & below)] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, c0, 0 };
			if (h1 && c1 < 0)
				out[n + nl0 + __builtin_popcountll(l1 & below)] = { q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, c1, 0 };
			if (h0 && c0 >= 0)
				next[ni + __builtin_popcountll(i0 & below)] = c0;
			if (h1 && c1 >= 0)
				next[ni + ni0 + __builtin_popcountll(i1 & below)] = c1;
			n += nl;
			ni += nin;
		}
		nf = ni;
		int *t = front; front = next; next = t;
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	return n;
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
	// the lanes whose ray is occluded, as a lane mask in scalar registers (a per-lane flag
	// carried round the loops costs two vector instructions wherever it meets a mask)
	u64 found_m = 0;
	const u64 lanes = __builtin_amdgcn_ballot_w64(true);
	RT_STAT(unsigned st_mine = 0; unsigned st_union = 0);
	RT_STAT(const unsigned st_lanes = (unsigned)__builtin_popcountll(lanes));
	for (int c = 0; c < count; c++) {
		const ShaftEntry e = list[c];
		RT_EV(EV_LIST_BOX);
		float tn;
		u64 actm = __builtin_amdgcn_ballot_w64(packet_box_hit(pr, e.lox, e.loy, e.loz, e.hix, e.hiy, e.hiz, tnear, tfar, tn)) & ~found_m;
		RT_STAT(st_mine += (actm >> lane_id()) & 1u; st_union += actm ? 1u : 0u);
		if (actm == 0)
			continue;
		const int v = ~__builtin_amdgcn_readfirstlane(e.ref);
		const int first = v >> 2, ntri = (v & 3) + 1;
		for (int i = 0; i < ntri; i++) {
			const CONSTAS DevTriEdges &te = edges[first + i];
			float aa, bb, cc;
			double t;
			RT_STAT(if (lane_id() == (unsigned)__builtin_ctzll(lanes)) atomicAdd(&g_stat_tris, 1ull));
			const bool ok = triangle_intersect_packet(o, d, ld3(te.u), ld3(te.e1), ld3(te.e2), actm, wlo, whi, aa, bb, cc, t);
			(void)ok;          // t is NaN without a hit
			found_m |= __builtin_amdgcn_ballot_w64(t >= dnear) & __builtin_amdgcn_ballot_w64(t <= dfar);
			actm &= ~found_m;
			if (actm == 0)
				break;
		}
		if ((lanes & ~found_m) == 0)
			break;
	}
	const bool found = __builtin_amdgcn_inverse_ballot_w64(found_m);
#ifdef RT_TRAVERSAL_STATS
	{
		unsigned mx = st_mine, sm = st_mine;
		for (int off = 32; off > 0; off >>= 1) {
			mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
			sm += (unsigned)__shfl_xor((int)sm, off, 64);
		}
		if (lane_id() == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
			atomicAdd(&g_stat_list[0], 1ull);
			atomicAdd(&g_stat_list[1], (unsigned long long)count);
			atomicAdd(&g_stat_list[2], (unsigned long long)st_union);
			atomicAdd(&g_stat_list[3], (unsigned long long)mx);
			atomicAdd(&g_stat_list[4], (unsigned long long)sm);
			atomicAdd(&g_stat_list[5], (unsigned long long)st_lanes);
		}
	}
#endif
	return found;
}

// Builds the shaft of one area light for this wave: `olo`/`ohi` bound the hit
// points of the participating lanes (wave-uniform).  shading.c:22-47 defines the
// segments: origin = hit point + 1e-4 dir, end = origin + (float)|to_light| dir,
// the light point = position + alpha edge1 + beta edge2 with alpha, beta in [0, 2 - 1/n].
DEV ShaftState shaft_build(const DevScene &sc, CLight &light, int n, const float olo[3], const float ohi[3],
		int *wstack, ShaftEntry *lists, int stat_tag = 0)
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
	// the lists of a wave share one pool of entries: a list starts where the one before it ends
	int slot = 0, used = 0;
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
		const int room = RT_SHAFT_POOL - used < 254 ? RT_SHAFT_POOL - used : 254;
		const int n_leaves = shaft_collect(sc, in_const(sc.meshes)[s.mesh], model, wstack, lists + used, room);
		RT_STAT(if (counter) atomicAdd(&g_stat_shaft[n_leaves >= 0 ? n_leaves : RT_SHAFT_CAP + 1], 1ull));
		RT_STAT(if (counter && n_leaves < 0) atomicAdd(&g_stat_shaft[RT_SHAFT_CAP + 8 + stat_tag], 1ull));
		if (n_leaves >= 0) {
			used += n_leaves;
			ss.counts = (ss.counts & ~(0xffu << (8 * my_slot))) | ((unsigned)n_leaves << (8 * my_slot));
			ss.surfaces = (ss.surfaces & ~(0xffu << (8 * my_slot))) | ((unsigned)si << (8 * my_slot));
		}
	}
	return ss;
}

