/*
 * oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C CPU restatement of the reference's render path.  Every function
 * cites the reference lines it follows.  The arithmetic keeps the reference's
 * mixture of double vectors and float scalars, narrowing exactly where the
 * reference narrows; build with -ffp-contract=off and without -ffast-math so
 * that each operation rounds once, as in an IEEE build of the reference.
 *
 * Differences from the reference that are deliberate:
 *   - meshes are searched with a BVH instead of the kd-tree (mesh.c:350-451,
 *     ray.c:392-462).  The kd-tree's shape depends on uninitialised memory
 *     (mesh.c:320) and is not observable in the output; what is reproduced is
 *     its result: the smallest t among triangles that pass the barycentric
 *     tests inside the clipped [near, far] window.  Equal t: highest triangle
 *     index (the leaf loop's `<=`, ray.c:376, makes the later triangle win).
 *   - the "Unhandled case" trap of the cylinder (ray.c:263-268) returns a miss.
 *   - colour alpha is not carried (never output, ppm.c:22-24).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define TWO_PI 6.283185307179586476925287     /* cgmath.h:8 */
#define SQRT2 1.4142135623730950488016887242096980785696718753769480731766   /* cgmath.h:11 */
#define SQUARE(x) ((x) * (x))                 /* cgmath.h:18 */
#define MAXM(a, b) ((a) > (b) ? (a) : (b))    /* cgmath.h:15 */
#define MINM(a, b) ((a) < (b) ? (a) : (b))    /* cgmath.h:16 */

typedef struct V3 { double x, y, z; } V3;
typedef struct Col { float r, g, b; } Col;
typedef struct Ray { V3 origin, direction; float near, far; } Ray;   /* ray.h:7-12 */
typedef struct Hit {                                                    /* ray.h:14-19 */
	int surface;
	int triangle;
	V3 position, normal;
	double t;
} Hit;

/* ------------------------------------------------------------------ vectors */

static V3 vadd(V3 a, V3 b) { return (V3) { a.x + b.x, a.y + b.y, a.z + b.z }; }   /* vector.c:11 */
static V3 vsub(V3 a, V3 b) { return (V3) { a.x - b.x, a.y - b.y, a.z - b.z }; }   /* vector.c:21 */
static double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      /* vector.c:31 */
static double vlen(V3 a) { return sqrt(vdot(a, a)); }                             /* vector.c:36 */
static V3 vscale(double s, V3 a) { return (V3) { a.x * s, a.y * s, a.z * s }; }   /* vector.c:41 */
static V3 vnorm(V3 a) { return vscale(1 / vlen(a), a); }                          /* vector.c:51 */
static V3 vcross(V3 a, V3 b)                                                      /* vector.c:56 */
{
	return (V3) { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
static V3 vreflect(V3 d, V3 n) { return vadd(d, vscale(-2 * vdot(d, n), n)); }    /* vector.c:77 */

#define M4(m, i, j) (m)[4 * (j) + (i)]

static V3 xform_homo(const double *m, V3 v)     /* matrix.c:61-81, vector.c:95-104 */
{
	const double v0 = v.x, v1 = v.y, v2 = v.z, v3 = 1.0;
	double x = M4(m, 0, 0) * v0 + M4(m, 0, 1) * v1 + M4(m, 0, 2) * v2 + M4(m, 0, 3) * v3;
	double y = M4(m, 1, 0) * v0 + M4(m, 1, 1) * v1 + M4(m, 1, 2) * v2 + M4(m, 1, 3) * v3;
	double z = M4(m, 2, 0) * v0 + M4(m, 2, 1) * v1 + M4(m, 2, 2) * v2 + M4(m, 2, 3) * v3;
	double w = M4(m, 3, 0) * v0 + M4(m, 3, 1) * v1 + M4(m, 3, 2) * v2 + M4(m, 3, 3) * v3;
	return (V3) { x / w, y / w, z / w };
}

static V3 xform_hetero(const double *m, V3 v)   /* matrix.c:83-96 */
{
	const double v0 = v.x, v1 = v.y, v2 = v.z;
	V3 out;
	out.x = M4(m, 0, 0) * v0 + M4(m, 0, 1) * v1 + M4(m, 0, 2) * v2 + M4(m, 0, 3) * 0;
	out.y = M4(m, 1, 0) * v0 + M4(m, 1, 1) * v1 + M4(m, 1, 2) * v2 + M4(m, 1, 3) * 0;
	out.z = M4(m, 2, 0) * v0 + M4(m, 2, 1) * v1 + M4(m, 2, 2) * v2 + M4(m, 2, 3) * 0;
	return out;
}

/* --------------------------------------------------------------- colours */

static const Col BLACK = { 0.0f, 0.0f, 0.0f };
static Col cadd(Col a, Col b) { return (Col) { a.r + b.r, a.g + b.g, a.b + b.b }; }       /* colour.c:8 */
static Col cmul(Col a, Col b) { return (Col) { a.r * b.r, a.g * b.g, a.b * b.b }; }       /* colour.c:19 */
static Col cscale(float s, Col a) { return (Col) { s * a.r, s * a.g, s * a.b }; }          /* colour.c:30 */
static Col col3(const float *c) { return (Col) { c[0], c[1], c[2] }; }

/* ------------------------------------------------------------------- BVH */

typedef struct BvhNode {
	double lo[3], hi[3];
	int left, right;       /* children, or -1 */
	int first, count;      /* leaf range in `order` */
} BvhNode;

typedef struct OMesh {
	int nv, nn, nt;
	const double *vertex, *normal;
	const int32_t *tri;
	BvhNode *nodes;
	int nnodes;
	int *order;
} OMesh;

struct oracle_scene {
	void *blob;
	const rtamd_blob_header *h;
	const rtamd_light *lights;
	const rtamd_material *mats;
	const rtamd_surface *surfs;
	const rtamd_cubemap *cube;
	OMesh *meshes;
};

static void tri_bounds(const OMesh *m, int t, double lo[3], double hi[3])
{
	for (int a = 0; a < 3; a++) {
		lo[a] = HUGE_VAL;
		hi[a] = -HUGE_VAL;
	}
	for (int c = 0; c < 3; c++) {
		const double *v = m->vertex + 3 * m->tri[6 * t + c];
		for (int a = 0; a < 3; a++) {
			if (v[a] < lo[a]) lo[a] = v[a];
			if (v[a] > hi[a]) hi[a] = v[a];
		}
	}
}

static int g_sort_axis;
static const OMesh *g_sort_mesh;

static double tri_centroid(const OMesh *m, int t, int axis)
{
	double lo[3], hi[3];
	tri_bounds(m, t, lo, hi);
	return 0.5 * (lo[axis] + hi[axis]);
}

static int cmp_centroid(const void *a, const void *b)
{
	double ca = tri_centroid(g_sort_mesh, *(const int *)a, g_sort_axis);
	double cb = tri_centroid(g_sort_mesh, *(const int *)b, g_sort_axis);
	return (ca > cb) - (ca < cb);
}

static int bvh_build(OMesh *m, int first, int count)
{
	int id = m->nnodes++;
	BvhNode *n = &m->nodes[id];
	for (int a = 0; a < 3; a++) {
		n->lo[a] = HUGE_VAL;
		n->hi[a] = -HUGE_VAL;
	}
	for (int i = first; i < first + count; i++) {
		double lo[3], hi[3];
		tri_bounds(m, m->order[i], lo, hi);
		for (int a = 0; a < 3; a++) {
			if (lo[a] < n->lo[a]) n->lo[a] = lo[a];
			if (hi[a] > n->hi[a]) n->hi[a] = hi[a];
		}
	}
	n->left = n->right = -1;
	n->first = first;
	n->count = count;
	if (count <= 4)
		return id;
	int axis = 0;
	double ext = n->hi[0] - n->lo[0];
	for (int a = 1; a < 3; a++)
		if (n->hi[a] - n->lo[a] > ext) {
			ext = n->hi[a] - n->lo[a];
			axis = a;
		}
	g_sort_axis = axis;
	g_sort_mesh = m;
	qsort(m->order + first, (size_t)count, sizeof(int), cmp_centroid);
	int half = count / 2;
	int l = bvh_build(m, first, half);
	int r = bvh_build(m, first + half, count - half);
	m->nodes[id].left = l;
	m->nodes[id].right = r;
	return id;
}

/* Conservative slab test: true unless the box certainly lies outside
 * [tlo, thi] along the ray; any NaN counts as a possible hit. */
static int box_maybe_hit(const BvhNode *n, V3 o, V3 d, double tlo, double thi)
{
	const double oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
	double t0 = tlo, t1 = thi;
	for (int a = 0; a < 3; a++) {
		double span = n->hi[a] - n->lo[a];
		double pad = 1e-9 * (fabs(n->hi[a]) + fabs(n->lo[a]) + span) + 1e-12;
		double lo = n->lo[a] - pad, hi = n->hi[a] + pad;
		if (dd[a] == 0.0) {
			if (oo[a] < lo || oo[a] > hi)
				return 0;
			continue;
		}
		double ta = (lo - oo[a]) / dd[a], tb = (hi - oo[a]) / dd[a];
		if (ta > tb) { double s = ta; ta = tb; tb = s; }
		ta -= fabs(ta) * 1e-12;
		tb += fabs(tb) * 1e-12;
		if (ta > t0) t0 = ta;
		if (tb < t1) t1 = tb;
	}
	return !(t0 > t1);
}

struct TriangleHit { double t; float a, b, c; int triangle; };   /* ray.c:6-12 */

/* ray.c:327-358 */
static int ray_triangle_intersect(V3 origin, V3 direction, V3 u, V3 v, V3 w,
		struct TriangleHit *tri_hit)
{
	V3 edge1, edge2, tvec, pvec, qvec;
	float det;
	float aa, bb, cc;

	edge1 = vsub(v, u);
	edge2 = vsub(w, u);

	pvec = vcross(direction, edge2);
	det = vdot(edge1, pvec);

	tvec = vsub(origin, u);
	bb = vdot(tvec, pvec) / det;
	if (bb < 0.0 || bb > 1.0)
		return 0;

	qvec = vcross(tvec, edge1);
	cc = vdot(direction, qvec) / det;
	if (cc < 0.0 || cc + bb > 1.0)
		return 0;

	aa = 1.0 - bb - cc;

	tri_hit->a = aa;
	tri_hit->b = bb;
	tri_hit->c = cc;
	tri_hit->t = vdot(edge2, qvec) / det;
	return 1;
}

static V3 mesh_vertex(const OMesh *m, int t, int c)
{
	const double *v = m->vertex + 3 * m->tri[6 * t + c];
	return (V3) { v[0], v[1], v[2] };
}

static V3 mesh_normal(const OMesh *m, int t, int c)
{
	const double *v = m->normal + 3 * m->tri[6 * t + 3 + c];
	return (V3) { v[0], v[1], v[2] };
}

/* The acceptance rule of ray.c:376 with tie -> highest index. */
static void consider_triangle(const OMesh *m, int t, const Ray *ray, struct TriangleHit *best)
{
	struct TriangleHit nhit;
	if (!ray_triangle_intersect(ray->origin, ray->direction, mesh_vertex(m, t, 0),
				mesh_vertex(m, t, 1), mesh_vertex(m, t, 2), &nhit))
		return;
	if (nhit.t >= ray->near && nhit.t <= ray->far &&
			(nhit.t < best->t || (nhit.t == best->t && t > best->triangle))) {
		*best = nhit;
		best->triangle = t;
	}
}

/* Effective result of ray.c:360-462 (see file header). */
static int mesh_nearest(const OMesh *m, const Ray *ray, struct TriangleHit *out, int brute)
{
	struct TriangleHit best;
	best.t = HUGE_VAL;
	best.triangle = -1;
	if (brute) {
		for (int t = 0; t < m->nt; t++)
			consider_triangle(m, t, ray, &best);
	} else if (m->nnodes > 0) {
		int stack[128], sp = 0;
		stack[sp++] = 0;
		while (sp) {
			const BvhNode *n = &m->nodes[stack[--sp]];
			double thi = best.t < ray->far ? best.t : ray->far;
			if (!box_maybe_hit(n, ray->origin, ray->direction, ray->near, thi))
				continue;
			if (n->left < 0) {
				for (int i = n->first; i < n->first + n->count; i++)
					consider_triangle(m, m->order[i], ray, &best);
			} else {
				stack[sp++] = n->left;
				stack[sp++] = n->right;
			}
		}
	}
	if (best.triangle < 0)
		return 0;
	*out = best;
	return 1;
}

/* ------------------------------------------------------- scene lifetime */

oracle_scene *oracle_scene_create(const void *blob, size_t blob_size)
{
	if (!blob || blob_size < sizeof(rtamd_blob_header))
		return NULL;
	oracle_scene *s = calloc(1, sizeof *s);
	s->blob = malloc(blob_size);
	memcpy(s->blob, blob, blob_size);
	const uint8_t *base = s->blob;
	s->h = (const rtamd_blob_header *)base;
	if (s->h->magic != RTAMD_BLOB_MAGIC || s->h->total_size != blob_size) {
		free(s->blob);
		free(s);
		return NULL;
	}
	s->lights = (const rtamd_light *)(base + s->h->off_lights);
	s->mats = (const rtamd_material *)(base + s->h->off_materials);
	s->surfs = (const rtamd_surface *)(base + s->h->off_surfaces);
	s->cube = s->h->has_cubemap ? (const rtamd_cubemap *)(base + s->h->off_cubemap) : NULL;
	const rtamd_mesh *md = (const rtamd_mesh *)(base + s->h->off_meshes);
	s->meshes = calloc((size_t)(s->h->num_meshes ? s->h->num_meshes : 1), sizeof(OMesh));
	for (int i = 0; i < s->h->num_meshes; i++) {
		OMesh *m = &s->meshes[i];
		m->nv = md[i].num_vertices;
		m->nn = md[i].num_normals;
		m->nt = md[i].num_triangles;
		m->vertex = (const double *)(base + md[i].off_vertex);
		m->normal = (const double *)(base + md[i].off_normal);
		m->tri = (const int32_t *)(base + md[i].off_triangle);
		m->nodes = calloc((size_t)(2 * m->nt + 1), sizeof(BvhNode));
		m->order = calloc((size_t)(m->nt ? m->nt : 1), sizeof(int));
		for (int t = 0; t < m->nt; t++)
			m->order[t] = t;
		m->nnodes = 0;
		if (m->nt > 0)
			bvh_build(m, 0, m->nt);
	}
	return s;
}

void oracle_scene_destroy(oracle_scene *s)
{
	if (!s)
		return;
	for (int i = 0; i < s->h->num_meshes; i++) {
		free(s->meshes[i].nodes);
		free(s->meshes[i].order);
	}
	free(s->meshes);
	free(s->blob);
	free(s);
}

/* ----------------------------------------------------------- RNG sources */

/* glibc TYPE_3 rand(): see the stream definition in the header comment of
 * rtamd_glibc_rand_fill; restated here independently of the product code. */
typedef struct Glibc { uint32_t r[34]; int idx; } Glibc;

static void glibc_seed(Glibc *g, uint32_t seed)
{
	int32_t r[34];
	r[0] = seed ? (int32_t)seed : 1;
	for (int i = 1; i < 31; i++) {
		int64_t w = (16807LL * r[i - 1]) % 2147483647;
		if (w < 0)
			w += 2147483647;
		r[i] = (int32_t)w;
	}
	for (int i = 31; i < 34; i++)
		r[i] = r[i - 31];
	for (int i = 0; i < 34; i++)
		g->r[i] = (uint32_t)r[i];
	g->idx = 0;
	for (int i = 0; i < 310; i++) {
		g->r[g->idx] = g->r[(g->idx + 3) % 34] + g->r[(g->idx + 31) % 34];
		g->idx = (g->idx + 1) % 34;
	}
}

static int32_t glibc_rand(Glibc *g)
{
	uint32_t v = g->r[(g->idx + 3) % 34] + g->r[(g->idx + 31) % 34];
	g->r[g->idx] = v;
	g->idx = (g->idx + 1) % 34;
	return (int32_t)(v >> 1);
}

static uint32_t mix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x7feb352du;
	h ^= h >> 15;
	h *= 0x846ca68bu;
	h ^= h >> 16;
	return h;
}

static uint32_t hash2(uint32_t a, uint32_t b)
{
	return mix32(a ^ (mix32(b) + 0x9e3779b9u + (a << 6) + (a >> 2)));
}

uint32_t oracle_keyed_key0(uint32_t seed, uint32_t pixel, uint32_t sample)
{
	return hash2(hash2(seed, pixel), sample);
}

uint32_t oracle_keyed_child(uint32_t key, uint32_t child)
{
	return hash2(key ^ 0xa511e9b3u, child + 1u);
}

uint32_t oracle_keyed_draw(uint32_t key, uint32_t purpose, uint32_t index)
{
	return hash2(hash2(key, purpose), index) >> 1;
}

enum { PURPOSE_AA = 0, PURPOSE_LIGHT0 = 1, PURPOSE_GLOSSY = 16 };

/* Per-thread render context. */
typedef struct Ctx {
	const oracle_scene *s;
	int rng_mode;
	Glibc *glibc;
	oracle_stats st;
	double *ray_log;          /* optional: 8 doubles per ray_intersect call */
	size_t ray_log_cap, ray_log_n;
} Ctx;

/* drand (ray.c:14-17): rand()/RAND_MAX, or the keyed equivalent. */
static double drand_src(Ctx *c, uint32_t key, uint32_t purpose, uint32_t index)
{
	c->st.draws++;
	if (c->rng_mode == 0)
		return glibc_rand(c->glibc) / ((double)2147483647);
	/* keyed sequence: scaled by the double nearest to 1/RAND_MAX (same definition
	 * as the HIP kernels) */
	return oracle_keyed_draw(key, purpose, index) * 4.6566128752457969e-10;
}

/* ------------------------------------------------------------ camera rays */

/* ray.c:19-44 */
static Ray cam_ray_internal(const oracle_scene *s, int i, int j, float offx, float offy,
		double near)
{
	Ray r;
	float d, u, v;
	double bottom, left, width, height;
	const rtamd_blob_header *h = s->h;
	const int nx = h->width, ny = h->height;
	V3 cu = { h->cam_u[0], h->cam_u[1], h->cam_u[2] };
	V3 cv = { h->cam_v[0], h->cam_v[1], h->cam_v[2] };
	V3 cw = { h->cam_w[0], h->cam_w[1], h->cam_w[2] };

	width = 2 * near * tan(h->cam_fov * TWO_PI / 360. / 2.);
	left = -width / 2;
	height = width * ny / (double)nx;
	bottom = -height / 2;

	d = near;
	u = left + width * (i + offx) / nx;
	v = bottom + height * (j + offy) / ny;
	r.origin = (V3) { h->cam_position[0], h->cam_position[1], h->cam_position[2] };
	r.direction = vnorm(vadd(vscale(-d, cw), vadd(vscale(u, cu), vscale(v, cv))));
	r.near = 0;
	r.far = HUGE_VAL;
	return r;
}

/* ray.c:47-59 */
static Ray camera_ray_aa(Ctx *c, int i, int j, int sample, double near, uint32_t key)
{
	float offx, offy;
	int p, q;
	const int aa = c->s->h->aa_samples;
	const float n = (float)aa;

	p = sample % aa;
	q = sample / aa;
	offx = (p + drand_src(c, key, PURPOSE_AA, 0)) / n;
	offy = (q + drand_src(c, key, PURPOSE_AA, 1)) / n;
	return cam_ray_internal(c->s, i, j, offx, offy, near);
}

/* ------------------------------------------------------------ intersectors */

/* ray.c:69-97 */
static int ray_plane_intersect(Ray ray, const rtamd_surface *sf, float t[1], V3 normal[1])
{
	float test_t, alpha, beta, det;
	V3 a = { sf->edge1[0], sf->edge1[1], sf->edge1[2] };
	V3 b = { sf->edge2[0], sf->edge2[1], sf->edge2[2] };
	V3 d = ray.direction, o = ray.origin;
	V3 n, axn, bxn, pos;

	n = vcross(a, b);
	test_t = -vdot(o, n) / vdot(d, n);
	pos = vadd(o, vscale(test_t, d));

	axn = vcross(a, n);
	bxn = vcross(b, n);
	det = vdot(a, bxn);

	alpha = vdot(pos, bxn) / det;
	beta = -vdot(pos, axn) / det;

	if (alpha < 0 || alpha > 1 || beta < 0 || beta > 1)
		return 0;

	t[0] = test_t;
	if (vdot(d, n) < 0)
		normal[0] = n;
	else
		normal[0] = vscale(-1, n);
	return 1;
}

/* ray.c:99-120 */
static int ray_disk_intersect(Ray ray, const rtamd_surface *sf, float t[1], V3 normal[1])
{
	double tt, xx, yy;
	V3 o = ray.origin, d = ray.direction;
	const float radius = sf->radius;

	if (d.z == 0)
		return 0;

	tt = -o.z / d.z;
	xx = o.x + tt * d.x;
	yy = o.y + tt * d.y;
	if (SQUARE(xx) + SQUARE(yy) > SQUARE(radius))
		return 0;

	t[0] = tt;
	if (o.z + ray.near * d.z >= 0)
		normal[0] = (V3) { 0, 0, 1 };
	else
		normal[0] = (V3) { 0, 0, -1 };
	return 1;
}

/* ray.c:122-151 */
static int ray_sphere_intersect(Ray r, const rtamd_surface *sf, float t[2], V3 normal[2])
{
	V3 v;
	float dd, vd, vv, discriminant;
	float radius = sf->radius;

	v = r.origin;
	vv = vdot(v, v);
	vd = vdot(v, r.direction);
	dd = vdot(r.direction, r.direction);

	discriminant = vd * vd - dd * (vv - radius * radius);
	if (discriminant < 0)
		return 0;
	else if (discriminant == 0) {
		t[0] = -vd / dd;
		normal[0] = vnorm(vadd(r.origin, vscale(t[0], r.direction)));
		return 1;
	} else {
		t[0] = (-vd - sqrtf(discriminant)) / dd;
		t[1] = (-vd + sqrtf(discriminant)) / dd;
		normal[0] = vadd(r.origin, vscale(t[0], r.direction));
		normal[1] = vadd(r.origin, vscale(t[1], r.direction));
		return 2;
	}
}

/* ray.c:153-271 */
static int ray_cylinder_intersect(Ray ray, const rtamd_surface *sf, float t[2], V3 normal[2])
{
	const float height = sf->height, radius = sf->radius;
	const V3 o = ray.origin, d = ray.direction;
	float a, b, c, disc, z0, z1;
	int capped = sf->capped;

	a = SQUARE(d.x) + SQUARE(d.y);
	b = 2 * (o.x * d.x + o.y * d.y);
	c = SQUARE(o.x) + SQUARE(o.y) - SQUARE(radius);

	disc = b * b - 4 * a * c;
	if (disc < 0)
		return 0;

	t[0] = (-b - sqrtf(disc)) / (2 * a);
	normal[0].x = (o.x + t[0] * d.x) / radius;
	normal[0].y = (o.y + t[0] * d.y) / radius;
	normal[0].z = 0;
	t[1] = (-b + sqrtf(disc)) / (2 * a);
	normal[1].x = (o.x + t[1] * d.x) / radius;
	normal[1].y = (o.y + t[1] * d.y) / radius;
	normal[1].z = 0;

	z0 = o.z + t[0] * d.z;
	z1 = o.z + t[1] * d.z;
	if ((z0 < 0 && z1 < 0) || (z0 > height && z1 > height))
		return 0;
	if ((z0 < 0 && z1 > height) || (z0 > height && z1 < 0)) {
		if (!capped)
			return 0;
		t[0] = -o.z / d.z;
		normal[0] = (V3) { 0, 0, -1 };
		t[1] = (height - o.z) / d.z;
		normal[1] = (V3) { 0, 0, 1 };
	} else if (z0 >= 0 && z0 <= height && z1 > height) {
		if (capped) {
			t[1] = (height - o.z) / d.z;
			normal[1] = (V3) { 0, 0, 1 };
		} else {
			t[1] = t[0];
			normal[1] = normal[0];
		}
	} else if (z0 >= 0 && z0 <= height && z1 < 0) {
		if (capped) {
			t[1] = -o.z / d.z;
			normal[1] = (V3) { 0, 0, -1 };
		} else {
			t[1] = t[0];
			normal[1] = normal[0];
		}
	} else if (z1 >= 0 && z1 <= height && z0 > height) {
		if (capped) {
			t[0] = (height - o.z) / d.z;
			normal[0] = (V3) { 0, 0, 1 };
		} else {
			t[0] = t[1];
			normal[0] = normal[1];
		}
	} else if (z1 >= 0 && z1 <= height && z0 < 0) {
		if (capped) {
			t[0] = -o.z / d.z;
			normal[0] = (V3) { 0, 0, -1 };
		} else {
			t[0] = t[1];
			normal[0] = normal[1];
		}
	} else if (z0 >= 0 && z0 <= height && z1 >= 0 && z1 <= height) {
		/* nothing to fix */
	} else {
		return 0;    /* the reference traps here (ray.c:263-268) */
	}
	return 2;
}

/* ray.c:273-324 */
static int ray_cone_intersect(Ray ray, const rtamd_surface *sf, float t[2], V3 normal[2])
{
	float dx, dy, dz, ox, oy, oz, R, h;
	float a, b, c, disc;
	float z0, z1;

	dx = ray.direction.x;
	dy = ray.direction.y;
	dz = ray.direction.z;
	ox = ray.origin.x;
	oy = ray.origin.y;
	oz = ray.origin.z;
	R = sf->radius;
	h = sf->height;

	a = SQUARE(dx) + SQUARE(dy) - SQUARE(R / h * dz);
	b = 2 * (ox * dx + oy * dy + SQUARE(R / h) * (-oz * dz + h * dz));
	c = SQUARE(ox) + SQUARE(oy) - SQUARE(R / h) * (SQUARE(h) - 2 * h * oz + SQUARE(oz));

	disc = b * b - 4 * a * c;
	if (disc < 0)
		return 0;

	t[0] = (-b - sqrtf(disc)) / (2 * a);
	t[1] = (-b + sqrtf(disc)) / (2 * a);

	z0 = oz + t[0] * dz;
	z1 = oz + t[1] * dz;

	if ((z0 > h || z0 < 0) && (z1 > h || z1 < 0))
		return 0;
	else if (z0 > h || z0 < 0)
		t[0] = t[1];
	else if (z1 > h || z1 < 0)
		t[1] = t[0];

	normal[0].x = h / sqrtf(h * h + R * R) * (ox + t[0] * dx) /
			sqrtf(SQUARE(ox + t[0] * dx) + SQUARE(oy + t[0] * dy));
	normal[0].y = h / sqrtf(h * h + R * R) * (oy + t[0] * dy) /
			sqrtf(SQUARE(ox + t[0] * dx) + SQUARE(oy + t[0] * dy));
	normal[0].z = R / sqrtf(h * h + R * R);
	normal[1] = normal[0];     /* never read: the function reports one hit (ray.c:323) */
	return 1;
}

/* ray.c:464-481 */
static int ray_mesh_intersect(Ray ray, const OMesh *mesh, float *t, V3 *normal, int *triangle)
{
	struct TriangleHit tri_hit;
	if (mesh_nearest(mesh, &ray, &tri_hit, 0)) {
		int tr = tri_hit.triangle;
		*t = tri_hit.t;
		*normal = vadd(vadd(vscale(tri_hit.a, mesh_normal(mesh, tr, 0)),
					vscale(tri_hit.b, mesh_normal(mesh, tr, 1))),
				vscale(tri_hit.c, mesh_normal(mesh, tr, 2)));
		*triangle = tr;
		return 1;
	}
	return 0;
}

/* ray.c:483-578 */
static int ray_surface_intersect(Ctx *c, Ray ray, int si, Hit *hit)
{
	const rtamd_surface *surf = &c->s->surfs[si];
	float ts[2] = { -HUGE_VAL, -HUGE_VAL }, t;
	V3 tnormals[2] = { { 0, 0, 0 }, { 0, 0, 0 } }, tnormal;
	Ray tray;
	int hits, triangle = -1;

	c->st.surface_probes++;
	tray.origin = xform_homo(surf->world_to_model, ray.origin);
	tray.direction = xform_hetero(surf->world_to_model, ray.direction);
	tray.near = ray.near;
	tray.far = ray.far;

	switch (surf->shape_type) {
	case RTAMD_SHAPE_PLANE:
		hits = ray_plane_intersect(tray, surf, ts, tnormals);
		break;
	case RTAMD_SHAPE_DISK:
		hits = ray_disk_intersect(tray, surf, ts, tnormals);
		break;
	case RTAMD_SHAPE_SPHERE:
		hits = ray_sphere_intersect(tray, surf, ts, tnormals);
		break;
	case RTAMD_SHAPE_CYLINDER:
		hits = ray_cylinder_intersect(tray, surf, ts, tnormals);
		break;
	case RTAMD_SHAPE_CONE:
		hits = ray_cone_intersect(tray, surf, ts, tnormals);
		break;
	case RTAMD_SHAPE_MESH: {
		const OMesh *mesh = &c->s->meshes[surf->mesh];
		int levels = 0;
		while ((4 << levels) < mesh->nt)
			levels++;
		c->st.mesh_probes++;
		c->st.mesh_probe_levels += (uint64_t)levels;
		hits = ray_mesh_intersect(tray, mesh, ts, tnormals, &triangle);
		break;
	}
	default:
		return 0;
	}

	if (hits == 0)
		return 0;
	if (hits == 1) {
		if (ts[0] < ray.near || ts[0] > ray.far)
			return 0;
		t = ts[0];
		tnormal = tnormals[0];
	} else {
		int t0_ok = ts[0] >= ray.near && ts[0] <= ray.far;
		int t1_ok = ts[1] >= ray.near && ts[1] <= ray.far;
		if (!t0_ok && !t1_ok)
			return 0;
		else if (t0_ok && !t1_ok) {
			t = ts[0];
			tnormal = tnormals[0];
		} else if (!t0_ok && t1_ok) {
			t = ts[1];
			tnormal = tnormals[1];
		} else if (ts[0] < ts[1]) {
			t = ts[0];
			tnormal = tnormals[0];
		} else {
			t = ts[1];
			tnormal = tnormals[1];
		}
	}

	hit->t = t;
	hit->triangle = triangle;
	hit->position = vadd(ray.origin, vscale(t, ray.direction));
	/* normal matrix = transpose(world_to_model), ray.c:492-493,576 */
	{
		const double *w = surf->world_to_model;
		const double n0 = tnormal.x, n1 = tnormal.y, n2 = tnormal.z;
		V3 n;
		n.x = M4(w, 0, 0) * n0 + M4(w, 1, 0) * n1 + M4(w, 2, 0) * n2 + M4(w, 3, 0) * 0;
		n.y = M4(w, 0, 1) * n0 + M4(w, 1, 1) * n1 + M4(w, 2, 1) * n2 + M4(w, 3, 1) * 0;
		n.z = M4(w, 0, 2) * n0 + M4(w, 1, 2) * n1 + M4(w, 2, 2) * n2 + M4(w, 3, 2) * 0;
		hit->normal = vnorm(n);
	}
	return 1;
}

/* ray.c:580-641 */
static int ray_bbox_test(Ray ray, const float *bbox, Ray *bray)
{
	float xmin = bbox[0], xmax = bbox[3];
	float ymin = bbox[1], ymax = bbox[4];
	float zmin = bbox[2], zmax = bbox[5];
	float txmin, txmax, tymin, tymax, tzmin, tzmax;
	float xd = ray.direction.x, yd = ray.direction.y, zd = ray.direction.z;
	float xe = ray.origin.x, ye = ray.origin.y, ze = ray.origin.z;
	float xa, ya, za;

	xa = 1.0 / xd;
	if (xa >= 0) {
		txmin = (xmin - xe) * xa;
		txmax = (xmax - xe) * xa;
	} else {
		txmin = (xmax - xe) * xa;
		txmax = (xmin - xe) * xa;
	}
	ya = 1.0 / yd;
	if (ya >= 0) {
		tymin = (ymin - ye) * ya;
		tymax = (ymax - ye) * ya;
	} else {
		tymin = (ymax - ye) * ya;
		tymax = (ymin - ye) * ya;
	}
	za = 1.0 / zd;
	if (za >= 0) {
		tzmin = (zmin - ze) * za;
		tzmax = (zmax - ze) * za;
	} else {
		tzmin = (zmax - ze) * za;
		tzmax = (zmin - ze) * za;
	}

	if ((txmin > tymax) || (tymin > txmax))
		return 0;
	if ((tymin > tzmax) || (tzmin > tymax))
		return 0;
	if ((tzmin > txmax) || (txmin > tzmax))
		return 0;

	bray->origin = ray.origin;
	bray->direction = ray.direction;
	bray->near = MAXM(txmin, MAXM(tymin, MAXM(tzmin, ray.near)));
	bray->far = MINM(txmax, MINM(tymax, MINM(tzmax, ray.far)));
	return 1;
}

/* ray.c:643-672 */
static int ray_intersect(Ctx *c, Ray ray, Hit *hit)
{
	Hit test_hit;
	if (c->ray_log && c->ray_log_n < c->ray_log_cap) {
		double *o = c->ray_log + 8 * c->ray_log_n++;
		o[0] = ray.origin.x; o[1] = ray.origin.y; o[2] = ray.origin.z;
		o[3] = ray.direction.x; o[4] = ray.direction.y; o[5] = ray.direction.z;
		o[6] = ray.near; o[7] = ray.far;
	}
	hit->surface = -1;
	hit->triangle = -1;
	hit->t = HUGE_VAL;
	for (int si = 0; si < c->s->h->num_surfaces; si++) {
		Ray bray;
		c->st.bbox_tests++;
		if (!ray_bbox_test(ray, c->s->surfs[si].bbox, &bray))
			continue;
		test_hit.surface = si;
		if (ray_surface_intersect(c, bray, si, &test_hit)) {
			if (hit->surface < 0 || test_hit.t < hit->t)
				*hit = test_hit;
		}
	}
	return hit->surface >= 0;
}

/* ---------------------------------------------------------------- texture */

/* texture.c:72-116 */
static Col texture_texel(const oracle_scene *s, int face, double u, double v)
{
	const int width = s->cube->width[face], height = s->cube->height[face];
	const float *buffer = (const float *)((const uint8_t *)s->blob + s->cube->off_face[face]);
	int x0, y0, x1, y1;
	double ualpha, ubeta, valpha, vbeta;
	Col caa, cab, cba, cbb;

	if (u == 1.0)
		u -= 1e-6;
	if (v == 1.0)
		v -= 1e-6;
	u = u - floor(u);
	v = v - floor(v);
	u *= width;
	v *= height;
	x0 = floor(u);
	y0 = floor(v);

	ualpha = u - x0;
	valpha = v - y0;
	ubeta = 1 - ualpha;
	vbeta = 1 - valpha;

	x1 = (x0 == width - 1) ? x0 : x0 + 1;
	y1 = (y0 == height - 1) ? y0 : y0 + 1;

	caa = col3(buffer + 4 * (width * y0 + x0));
	cab = col3(buffer + 4 * (width * y0 + x1));
	cba = col3(buffer + 4 * (width * y1 + x0));
	cbb = col3(buffer + 4 * (width * y1 + x1));

	return cadd(cscale(vbeta, cadd(cscale(ubeta, caa), cscale(ualpha, cab))),
			cscale(valpha, cadd(cscale(ubeta, cba), cscale(ualpha, cbb))));
}

/* texture.c:135-182; face numbering texture.h:6-10 */
static Col cubemap_colour(const oracle_scene *s, V3 d)
{
	int face = -1;
	float u = 0.0, v = 0.0;
	double dx = fabs(d.x), dy = fabs(d.y), dz = fabs(d.z);
	if (dx >= dy && dx >= dz) {
		if (d.x >= 0) {
			u = (1.0 + d.z / d.x) / 2.0;
			v = (1.0 + d.y / d.x) / 2.0;
			face = 1;
		} else {
			u = 1.0 - (1.0 - d.z / d.x) / 2.0;
			v = (1.0 - d.y / d.x) / 2.0;
			face = 0;
		}
	} else if (dy >= dz && dy >= dx) {
		if (d.y >= 0) {
			u = (1.0 + d.x / d.y) / 2.0;
			v = 1.0 - (1.0 - d.z / d.y) / 2.0;
			face = 3;
		} else {
			u = (1.0 - d.x / d.y) / 2.0;
			v = (1.0 + d.z / d.y) / 2.0;
			face = 2;
		}
	} else if (dz >= dx && dz >= dy) {
		if (d.z >= 0) {
			u = 1.0 - (1.0 + d.x / d.z) / 2.0;
			v = (1.0 + d.y / d.z) / 2.0;
			face = 5;
		} else {
			u = (1.0 - d.x / d.z) / 2.0;
			v = (1.0 - d.y / d.z) / 2.0;
			face = 4;
		}
	}
	if (face < 0)
		return BLACK;      /* NaN direction: the reference dereferences NULL here */
	return texture_texel(s, face, u, v);
}

/* --------------------------------------------------------------- shading */

/* lighting.c:7-19 */
static Col diff_colour(const rtamd_light *light, const rtamd_material *mat, V3 light_dir,
		V3 normal)
{
	float ndotv = MAXM(0, vdot(light_dir, normal));
	Col light_col = cscale(light->intensity, cscale(ndotv, col3(light->colour)));
	return cmul(col3(mat->diffuse), light_col);
}

/* lighting.c:21-34, called with (cam_dir, light_dir) in the (light_dir,
 * cam_dir) slots, shading.c:52 */
static Col spec_colour(const rtamd_light *light, const rtamd_material *mat, V3 light_dir,
		V3 cam_dir, V3 normal)
{
	V3 half = vnorm(vadd(light_dir, cam_dir));
	float hdotn = powf(MAXM(0, vdot(half, normal)), mat->shininess);
	Col light_col = cscale(light->intensity, cscale(hdotn, col3(light->colour)));
	return cmul(col3(mat->specular), light_col);
}

static Col ray_colour(Ctx *c, Ray ray, int depth, uint32_t key, int ray_kind);

/* shading.c:7-58 */
static Col hit_light_colour(Ctx *c, const Hit *hit, int li, V3 cam_dir, uint32_t key)
{
	const rtamd_light *light = &c->s->lights[li];
	const rtamd_material *mat = &c->s->mats[c->s->surfs[hit->surface].material];
	V3 normal = hit->normal;
	V3 light_dir, light_pos;
	V3 lpos = { light->position[0], light->position[1], light->position[2] };
	V3 e1 = { light->edge1[0], light->edge1[1], light->edge1[2] };
	V3 e2 = { light->edge2[0], light->edge2[1], light->edge2[2] };
	Col light_total = BLACK;
	int n = light->type == RTAMD_LIGHT_AREA ? c->s->h->shadow_samples : 1;

	for (int j = 0; j < SQUARE(n); j++) {
		Ray shadow_ray;
		Hit dummy;
		Col diff_col, spec_col;

		if (light->type == RTAMD_LIGHT_AREA) {
			int p, q;
			float alpha, beta;
			p = j % n;
			q = j / n;
			alpha = p / (float)n + drand_src(c, key, PURPOSE_LIGHT0 + (uint32_t)li, 2u * (uint32_t)j);
			beta = q / (float)n + drand_src(c, key, PURPOSE_LIGHT0 + (uint32_t)li, 2u * (uint32_t)j + 1u);
			light_pos = vadd(vadd(lpos, vscale(alpha, e1)), vscale(beta, e2));
		} else
			light_pos = lpos;

		light_dir = vnorm(vsub(light_pos, hit->position));

		shadow_ray.direction = light_dir;
		shadow_ray.origin = vadd(hit->position, vscale(1e-4, shadow_ray.direction));
		shadow_ray.near = 0;
		shadow_ray.far = vlen(vsub(light_pos, hit->position));
		c->st.rays_shadow++;
		if (ray_intersect(c, shadow_ray, &dummy))
			continue;

		c->st.light_samples++;
		diff_col = diff_colour(light, mat, light_dir, normal);
		spec_col = spec_colour(light, mat, cam_dir, light_dir, normal);
		light_total = cadd(light_total, cadd(diff_col, spec_col));
	}
	light_total = cscale(1.0 / SQUARE(n), light_total);
	return light_total;
}

/* shading.c:60-68 */
static V3 orthogonal(V3 v)
{
	const V3 n1 = { 1, 0, 0 }, n2 = { 0, 1, 0 };
	if (fabs(vdot(v, n1)) < vlen(v) / SQRT2)
		return vcross(v, n1);
	else
		return vcross(v, n2);
}

/* shading.c:70-112 */
static Col hit_reflection_colour(Ctx *c, const Hit *hit, Ray ray, int depth, uint32_t key)
{
	Col total;
	const rtamd_material *mat = &c->s->mats[c->s->surfs[hit->surface].material];
	Ray rray;

	if (mat->reflect <= 0.0)
		return BLACK;

	rray.direction = vreflect(ray.direction, hit->normal);
	rray.origin = vadd(hit->position, vscale(1e-2, rray.direction));
	rray.near = 0;
	rray.far = HUGE_VAL;

	if (mat->glossiness <= 0.0 || depth > 1)
		total = ray_colour(c, rray, depth + 1, oracle_keyed_child(key, 0), 2);
	else {
		const int rs = c->s->h->reflection_samples;
		total = BLACK;
		for (int i = 0; i < rs; i++) {
			Ray pray = rray;
			V3 a, b;
			pray.direction = vnorm(pray.direction);
			a = vnorm(orthogonal(pray.direction));
			b = vnorm(vcross(pray.direction, a));
			a = vscale(mat->glossiness * (2 * drand_src(c, key, PURPOSE_GLOSSY, 2u * (uint32_t)i) - 1), a);
			b = vscale(mat->glossiness * (2 * drand_src(c, key, PURPOSE_GLOSSY, 2u * (uint32_t)i + 1u) - 1), b);
			pray.direction = vadd(pray.direction, vadd(a, b));
			total = cadd(total, ray_colour(c, pray, depth + 1,
						oracle_keyed_child(key, (uint32_t)i), 2));
		}
		total = cscale(1. / rs, total);
	}
	return cmul(col3(mat->specular), cscale(mat->reflect, total));
}

/* shading.c:114-141.  ray_kind: 0 primary, 2 secondary (for the counters). */
static Col ray_colour(Ctx *c, Ray ray, int depth, uint32_t key, int ray_kind)
{
	Hit hit;
	Col total;
	V3 cam_dir = vnorm(vscale(-1, ray.direction));

	if (depth > c->s->h->max_reflections)
		return BLACK;

	if (ray_kind == 0)
		c->st.rays_primary++;
	else
		c->st.rays_secondary++;
	if (!ray_intersect(c, ray, &hit)) {
		if (c->s->cube) {
			c->st.env_lookups++;
			return cubemap_colour(c->s, ray.direction);
		} else
			return col3(c->s->h->background);
	}

	total = BLACK;
	for (int i = 0; i < c->s->h->num_lights; i++)
		total = cadd(total, hit_light_colour(c, &hit, i, cam_dir, key));
	total = cadd(total, hit_reflection_colour(c, &hit, ray, depth, key));
	return total;
}

/* ------------------------------------------------------------ entry points */

static void record_hit(rtamd_hit_record *rec, int did_hit, const Hit *hit)
{
	memset(rec, 0, sizeof *rec);
	rec->surface = -1;
	rec->triangle = -1;
	if (!did_hit)
		return;
	rec->surface = hit->surface;
	rec->triangle = hit->triangle;
	rec->t = (float)hit->t;
	rec->position[0] = hit->position.x;
	rec->position[1] = hit->position.y;
	rec->position[2] = hit->position.z;
	rec->normal[0] = hit->normal.x;
	rec->normal[1] = hit->normal.y;
	rec->normal[2] = hit->normal.z;
}

/* raytracer.c:32-54 */
static void pixel_colour(Ctx *c, int x, int y, uint32_t seed, float *out, rtamd_hit_record *hits)
{
	const rtamd_blob_header *h = c->s->h;
	Col col;
	Ray r;
	Hit hit;
	uint32_t pixel = (uint32_t)(h->width * y + x);

	if (h->antialiasing) {
		const int spp = SQUARE(h->aa_samples);
		col = BLACK;
		for (int k = 0; k < spp; k++) {
			uint32_t key = oracle_keyed_key0(seed, pixel, (uint32_t)k);
			r = camera_ray_aa(c, x, y, k, h->cam_near_plane, key);
			if (hits) {
				Ctx quiet = *c;     /* the extra probe must not disturb the counters */
				record_hit(&hits[k], ray_intersect(&quiet, r, &hit), &hit);
			}
			col = cadd(col, ray_colour(c, r, 0, key, 0));
		}
		col = cscale(1.0 / spp, col);
	} else {
		uint32_t key = oracle_keyed_key0(seed, pixel, 0);
		r = cam_ray_internal(c->s, x, y, 0.5, 0.5, 1);    /* ray.c:62-65, raytracer.c:49 */
		if (hits) {
			Ctx quiet = *c;
			record_hit(&hits[0], ray_intersect(&quiet, r, &hit), &hit);
		}
		col = ray_colour(c, r, 0, key, 0);
	}
	out[0] = col.r;
	out[1] = col.g;
	out[2] = col.b;
	out[3] = 0.0f;
}

typedef struct Job {
	Ctx ctx;
	uint32_t seed;
	int x0, y0, x1, y1;
	int row_first, row_step;
	float *colour;
	rtamd_hit_record *hits;
} Job;

static void *render_rows(void *arg)
{
	Job *job = arg;
	const rtamd_blob_header *h = job->ctx.s->h;
	const int spp = h->antialiasing ? SQUARE(h->aa_samples) : 1;
	for (int j = job->y0 + job->row_first; j < job->y1; j += job->row_step)
		for (int i = job->x0; i < job->x1; i++) {
			size_t local = (size_t)(j - job->y0) * (size_t)(job->x1 - job->x0) + (size_t)(i - job->x0);
			pixel_colour(&job->ctx, i, j, job->seed,
					job->colour + 4 * ((size_t)h->width * (size_t)j + (size_t)i),
					job->hits ? job->hits + local * (size_t)spp : NULL);
		}
	return NULL;
}

static void add_stats(oracle_stats *a, const oracle_stats *b)
{
	a->rays_primary += b->rays_primary;
	a->rays_shadow += b->rays_shadow;
	a->rays_secondary += b->rays_secondary;
	a->env_lookups += b->env_lookups;
	a->light_samples += b->light_samples;
	a->draws += b->draws;
	a->bbox_tests += b->bbox_tests;
	a->surface_probes += b->surface_probes;
	a->mesh_probes += b->mesh_probes;
	a->mesh_probe_levels += b->mesh_probe_levels;
}

int oracle_render(const oracle_scene *s, uint32_t seed, int rng_mode, int x0, int y0, int x1,
		int y1, float *colour, rtamd_hit_record *hits, oracle_stats *stats, int threads)
{
	if (!s || !colour)
		return -1;
	const rtamd_blob_header *h = s->h;
	const int windowed = !(x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0);
	if (!windowed) {
		x1 = h->width;
		y1 = h->height;
	}
	if (x0 < 0 || y0 < 0 || x1 > h->width || y1 > h->height || x0 >= x1 || y0 >= y1)
		return -1;
	const int spp = h->antialiasing ? SQUARE(h->aa_samples) : 1;
	oracle_stats total;
	memset(&total, 0, sizeof total);

	if (rng_mode == 0) {
		/* raytracer.c:74-84: one stream, raster order */
		Glibc g;
		Ctx ctx;
		memset(&ctx, 0, sizeof ctx);
		ctx.s = s;
		ctx.rng_mode = 0;
		ctx.glibc = &g;
		glibc_seed(&g, seed);
		for (int j = 0; j < h->height; j++)
			for (int i = 0; i < h->width; i++) {
				if (i < x0 || i >= x1 || j < y0 || j >= y1) {
					if (windowed && h->antialiasing)
						for (int d = 0; d < 2 * spp; d++)
							glibc_rand(&g);
					continue;
				}
				size_t local = (size_t)(j - y0) * (size_t)(x1 - x0) + (size_t)(i - x0);
				pixel_colour(&ctx, i, j, seed, colour + 4 * ((size_t)h->width * (size_t)j + (size_t)i),
						hits ? hits + local * (size_t)spp : NULL);
			}
		total = ctx.st;
	} else {
		if (threads < 1)
			threads = 1;
		if (threads > 256)
			threads = 256;
		Job *jobs = calloc((size_t)threads, sizeof *jobs);
		pthread_t *tid = calloc((size_t)threads, sizeof *tid);
		for (int t = 0; t < threads; t++) {
			jobs[t].ctx.s = s;
			jobs[t].ctx.rng_mode = 1;
			jobs[t].seed = seed;
			jobs[t].x0 = x0; jobs[t].y0 = y0; jobs[t].x1 = x1; jobs[t].y1 = y1;
			jobs[t].row_first = t;
			jobs[t].row_step = threads;
			jobs[t].colour = colour;
			jobs[t].hits = hits;
			if (threads == 1)
				render_rows(&jobs[t]);
			else
				pthread_create(&tid[t], NULL, render_rows, &jobs[t]);
		}
		for (int t = 0; t < threads; t++) {
			if (threads > 1)
				pthread_join(tid[t], NULL);
			add_stats(&total, &jobs[t].ctx.st);
		}
		free(jobs);
		free(tid);
	}
	if (stats)
		*stats = total;
	return 0;
}

int oracle_trace(const oracle_scene *s, const double *rays, size_t n, rtamd_hit_record *hits)
{
	if (!s || !rays || !hits)
		return -1;
	Ctx ctx;
	memset(&ctx, 0, sizeof ctx);
	ctx.s = s;
	for (size_t i = 0; i < n; i++) {
		const double *o = rays + 8 * i;
		Ray r;
		Hit hit;
		r.origin = (V3) { o[0], o[1], o[2] };
		r.direction = (V3) { o[3], o[4], o[5] };
		r.near = (float)o[6];
		r.far = (float)o[7];
		record_hit(&hits[i], ray_intersect(&ctx, r, &hit), &hit);
	}
	return 0;
}

int oracle_camera_rays(const oracle_scene *s, const int32_t *pixels, const float *offsets,
		double near_plane, size_t n, double *rays)
{
	if (!s || !pixels || !offsets || !rays)
		return -1;
	for (size_t i = 0; i < n; i++) {
		Ray r = cam_ray_internal(s, pixels[2 * i], pixels[2 * i + 1], offsets[2 * i],
				offsets[2 * i + 1], near_plane);
		double *o = rays + 8 * i;
		o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
		o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
		o[6] = r.near; o[7] = r.far;
	}
	return 0;
}

int oracle_mesh_bruteforce(const oracle_scene *s, int m, const double *ray, double *t_out)
{
	if (!s || m < 0 || m >= s->h->num_meshes)
		return -1;
	Ray r;
	struct TriangleHit th;
	r.origin = (V3) { ray[0], ray[1], ray[2] };
	r.direction = (V3) { ray[3], ray[4], ray[5] };
	r.near = (float)ray[6];
	r.far = (float)ray[7];
	if (!mesh_nearest(&s->meshes[m], &r, &th, 1))
		return -1;
	if (t_out)
		*t_out = th.t;
	return th.triangle;
}

/* Debugging aid: renders one pixel in serial-stream mode (draws of the pixels
 * before it are skipped, so the scene must have fixed consumption) and logs
 * every ray handed to ray_intersect.  Returns the number of rays logged. */
size_t oracle_debug_pixel_rays(const oracle_scene *s, uint32_t seed, int x, int y, double *rays,
		size_t cap, float *colour_out)
{
	const rtamd_blob_header *h = s->h;
	const int spp = h->antialiasing ? SQUARE(h->aa_samples) : 1;
	Glibc g;
	Ctx ctx;
	memset(&ctx, 0, sizeof ctx);
	ctx.s = s;
	ctx.glibc = &g;
	glibc_seed(&g, seed);
	if (h->antialiasing)
		for (long long d = 0; d < 2LL * spp * ((long long)h->width * y + x); d++)
			glibc_rand(&g);
	ctx.ray_log = rays;
	ctx.ray_log_cap = cap;
	float out[4];
	pixel_colour(&ctx, x, y, seed, out, NULL);
	if (colour_out)
		memcpy(colour_out, out, sizeof out);
	return ctx.ray_log_n;
}
