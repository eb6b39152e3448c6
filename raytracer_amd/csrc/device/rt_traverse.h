// BVH traversal of a mesh: the wave-wide packet walk (scalar node and triangle
// fetches, lane masks) and the per-lane walk.
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

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
		float tlo, float thi, float &tn, float &tf)
{
	float ax = __builtin_fmaf(lox, r.ix, r.kx), bx = __builtin_fmaf(hix, r.ix, r.kx);
	float ay = __builtin_fmaf(loy, r.iy, r.ky), by = __builtin_fmaf(hiy, r.iy, r.ky);
	float az = __builtin_fmaf(loz, r.iz, r.kz), bz = __builtin_fmaf(hiz, r.iz, r.kz);
	tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
			__builtin_fminf(az, bz)), tlo);
	tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
			__builtin_fmaxf(az, bz)), thi);
	return tn <= tf + r.slack;
}
DEV bool packet_box_hit(const PacketRay &r, float lox, float loy, float loz, float hix, float hiy, float hiz,
		float tlo, float thi, float &tn)
{
	float tf;
	return packet_box_hit(r, lox, loy, loz, hix, hiy, hiz, tlo, thi, tn, tf);
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

DEV bool triangle_intersect_packet(d3 o, d3 d, d3 u, d3 edge1, d3 edge2, u64 act, float wlo, float whi,
		float &aa, float &bb, float &cc, double &t)
{
	// (the lanes still in play are carried as a lane mask in scalar registers: a per-lane
	// flag that meets a ballot costs two vector instructions each time)
	// a lane that does not hit leaves t = NaN: the callers' window tests then fail by themselves
	t = __builtin_nan("");
	d3 pvec = vcross(d, edge2);
	float det = (float)vdot(edge1, pvec);
	d3 tvec = vsub(o, u);
	const double x1 = vdot(tvec, pvec);
	const float big = 3.0e38f;
	const float rdet = __builtin_amdgcn_rcpf(det);
	const float b_approx = (float)x1 * rdet;
	// (one ballot per comparison, combined as masks: a ballot of a compound condition is
	// compiled into a vector select and a second comparison)
#define LANES_WHERE(x) __builtin_amdgcn_ballot_w64(x)
	u64 maybe = act & ~((LANES_WHERE(b_approx < -1e-30f) & LANES_WHERE(b_approx > -big)) |
			(LANES_WHERE(b_approx > 1.0001f) & LANES_WHERE(b_approx < big)));
	if (maybe == 0)
		return false;
	RT_EV(EV_TRI_STAGE2);
	d3 qvec = vcross(tvec, edge1);
	const double x2 = vdot(d, qvec);
	const double x3 = vdot(edge2, qvec);
	const float c_approx = (float)x2 * rdet;
	const float s_approx = c_approx + b_approx;
	// relative error of t_approx below 2^-22 whenever it is finite: far inside the window's margin
	const float t_approx = (float)x3 * rdet;
	maybe &= ~((LANES_WHERE(c_approx < -1e-30f) & LANES_WHERE(c_approx > -big)) |
			(LANES_WHERE(s_approx > 1.0002f) & LANES_WHERE(s_approx < big) & LANES_WHERE(__builtin_fabsf(b_approx) < big)) |
			(LANES_WHERE(t_approx < wlo) & LANES_WHERE(t_approx > -big)) |
			(LANES_WHERE(t_approx > whi) & LANES_WHERE(t_approx < big)));
#undef LANES_WHERE
	if (maybe == 0)
		return false;
	RT_EV(EV_TRI_EXACT);
	bool ok = false;
	if (__builtin_amdgcn_inverse_ballot_w64(maybe)) {
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
	RT_STAT(unsigned walk_steps = 0);
	RT_STAT(if (ANYHIT) { RT_EV_ADD(EV_LANE_WALKS, 1); RT_EV_ADD(EV_LANE_WALK_LANES, __builtin_popcountll(__builtin_amdgcn_ballot_w64(true))); });
	for (;;) {
		RT_STAT(walk_steps++);
		RT_STAT(if (ANYHIT) { RT_EV_ADD(EV_LANE_STEPS, 1); RT_EV_ADD(EV_LANE_STEP_LANES, __builtin_popcountll(__builtin_amdgcn_ballot_w64(true))); });
		if (ref >= 0) {
			RT_STAT(atomicAdd(&g_stat_nodes, 1ull));
			const float4 q0 = nodes[4 * ref + 0], q1 = nodes[4 * ref + 1];
			const float4 q2 = nodes[4 * ref + 2], q3 = nodes[4 * ref + 3];
			float tn0, tn1, tf0, tf1;
			bool h0 = packet_box_hit(br, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tnear, limit, tn0, tf0);
			bool h1 = packet_box_hit(br, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tnear, limit, tn1, tf1);
			int c0 = as_i(q3.x), c1 = as_i(q3.y);
			if (h0 && h1) {
#if RT_ANYHIT_ORDER == 1
				// any hit will do: the child the ray spends more of its length in comes first
				bool swap = ANYHIT ? (tf1 - tn1) > (tf0 - tn0) : tn1 < tn0;
#elif RT_ANYHIT_ORDER == 2
				bool swap = ANYHIT ? tn1 > tn0 : tn1 < tn0;
#else
				bool swap = tn1 < tn0;
#endif
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
					if (ANYHIT) {
						RT_STAT(atomicAdd(&g_stat_walk[2 * (32 - __builtin_clz(walk_steps)) + 1 < 32 ? 2 * (32 - __builtin_clz(walk_steps)) + 1 : 31], 1ull));
						return true;
					}
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
	RT_STAT(if (ANYHIT) atomicAdd(&g_stat_walk[2 * (32 - __builtin_clz(walk_steps)) < 30 ? 2 * (32 - __builtin_clz(walk_steps)) : 30], 1ull));
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
			// front-to-back for the majority of the lanes; for shadow rays any order finds
			// the same answer, but lanes whose ray is blocked drop out sooner when the near
			// child comes first (7-reflections and 6-soft -2.5 %)
			const u64 pref1 = __builtin_amdgcn_ballot_w64(tn1 < tn0);
			const u64 both = m0 & m1;
			const u64 w1 = (both & pref1) | (m1 & ~m0), w0 = (both & ~pref1) | (m0 & ~m1);
			const bool swap = __builtin_popcountll(w1) > __builtin_popcountll(w0);
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
				const bool ok = triangle_intersect_packet(o, d, ld3(te.u), ld3(te.e1), ld3(te.e2), actm,
						wlo, window_hi(limit), aa, bb, cc, t);
				int orig = te.orig;
				(void)ok;      // t is NaN without a hit
				if (ANYHIT) {
					// any accepted triangle settles a lane (a lane takes part until it is settled,
					// so its best_t is still +inf): kept as a lane mask, nothing per lane
					donem |= __builtin_amdgcn_ballot_w64(t >= dnear) & __builtin_amdgcn_ballot_w64(t <= dfar) &
						__builtin_amdgcn_ballot_w64(t < __builtin_huge_val());
					actm &= ~donem;
					if (actm == 0)
						break;
				} else if (t >= dnear && t <= dfar && (t < best_t || (t == best_t && orig > best_orig))) {
					best_t = t;
					best_orig = orig;
					best_tri = first + i;
					ba = aa; bb_ = bb; bc = cc;
					found = true;
					// next float above or equal to t: boxes further than this cannot win
					float ft = (float)t;
					if ((double)ft < t)
						ft = next_up(ft);
					limit = ft < tfar ? ft : tfar;
				}
			}
		}
		TM2_END(TM_IN_LEAVES, t_leaves);
		// next stack entry that some lane still has to visit
		TM2_BEGIN(t_pop);
		for (;;) {
			if (sp == 0) {
				TM2_END(TM_IN_POP, t_pop);
				if (ANYHIT) {
					// the caller only asks whether: report the hit at the start of its window
					found = __builtin_amdgcn_inverse_ballot_w64(donem);
					if (found)
						best_t = dnear;
				}
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

