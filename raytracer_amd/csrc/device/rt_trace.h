// ray_intersect (ray.c:483-578, 643-672): nearest / any hit over the scene's
// surfaces, and the hit's position and normal.
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

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
	const SlabRay slab = slab_ray(ray);
	for (int si = next_wanted(0); si < sc.num_surfaces; si = next_wanted(si + 1)) {
		CSurface &s = in_const(sc.surfaces)[si];
		int shaft_count = 0xff, shaft_first = 0;
		if (masked) {
			// the lists lie one after the other in the wave's pool
			int first = 0;
			for (int k = 0; k < RT_SHAFT_LISTS; k++) {
				const int cnt = (int)((shaft.counts >> (8 * k)) & 0xffu);
				if ((int)((shaft.surfaces >> (8 * k)) & 0xffu) == si) {
					shaft_count = cnt;
					shaft_first = first;
				}
				if (cnt != 0xff)
					first += cnt;
			}
		}
		RT_EV(ANYHIT ? EV_BBOX_SHADOW : EV_BBOX_PRIMARY);
		float bnear, bfar;
		TM2_BEGIN(t_bbox);
		const bool in_box = surface_bbox_test(s.bbox, slab, ray.tnear, ray.tfar, bnear, bfar);
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
						shaft.lists + shaft_first, shaft_count) ? 1 : 0;
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

