// Surface slab test (ray.c:580-641), the five analytic primitives
// (ray.c:69-324) and the triangle test (ray.c:327-358).
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

// ---- ray.c:580-641 ------------------------------------------------------------
// What the slab test needs of the ray, the same for every surface: origin and
// direction narrowed to float and the reciprocals of ray.c:590,601,612.  Computed once
// per ray, not once per surface.
// The reference divides in double, 1.0 / (double)xd with xd a float, and narrows the
// quotient to float.  That is the correctly rounded float quotient 1.0f / xd: a double
// carries more than 2 x 24 + 2 bits, so rounding the double quotient a second time
// cannot land on the other side of a float (the classic result on double rounding of
// quotients; infinities, overflow and gradual underflow agree as well).  The float
// division below is compiled as the IEEE one (correctly rounded, denormals kept) and
// costs a third of the double sequence.
struct SlabRay {
	float xe, ye, ze;
	float xa, ya, za;
};

DEV SlabRay slab_ray(const Ray &ray)
{
	SlabRay r;
	r.xe = (float)ray.o.x; r.ye = (float)ray.o.y; r.ze = (float)ray.o.z;
	r.xa = 1.0f / (float)ray.d.x;
	r.ya = 1.0f / (float)ray.d.y;
	r.za = 1.0f / (float)ray.d.z;
	return r;
}

DEV bool surface_bbox_test(const CONSTAS float *bbox, const SlabRay &r, float tnear, float tfar, float &bnear, float &bfar)
{
	float xmin = bbox[0], xmax = bbox[3];
	float ymin = bbox[1], ymax = bbox[4];
	float zmin = bbox[2], zmax = bbox[5];
	float txmin, txmax, tymin, tymax, tzmin, tzmax;
	const float xe = r.xe, ye = r.ye, ze = r.ze;
	const float xa = r.xa, ya = r.ya, za = r.za;

	if (xa >= 0) { txmin = (xmin - xe) * xa; txmax = (xmax - xe) * xa; }
	else         { txmin = (xmax - xe) * xa; txmax = (xmin - xe) * xa; }
	if (ya >= 0) { tymin = (ymin - ye) * ya; tymax = (ymax - ye) * ya; }
	else         { tymin = (ymax - ye) * ya; tymax = (ymin - ye) * ya; }
	if (za >= 0) { tzmin = (zmin - ze) * za; tzmax = (zmax - ze) * za; }
	else         { tzmin = (zmax - ze) * za; tzmax = (zmin - ze) * za; }

	if ((txmin > tymax) || (tymin > txmax)) return false;
	if ((tymin > tzmax) || (tzmin > tymax)) return false;
	if ((tzmin > txmax) || (txmin > tzmax)) return false;

	// MAX/MIN macros of cgmath.h:15-16, written out so NaNs propagate the same way
	float m0 = tzmin > tnear ? tzmin : tnear;
	float m1 = tymin > m0 ? tymin : m0;
	bnear = txmin > m1 ? txmin : m1;
	float n0 = tzmax < tfar ? tzmax : tfar;
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

