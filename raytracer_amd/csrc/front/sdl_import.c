/* SDL scene import: the reference's sdl_load (scene.c:674-704) and everything
 * below it, producing one flat scene blob instead of the Sdl/Scene/Config
 * object graph.
 *
 * libxml2 with XML_PARSE_DTDLOAD (scene.c:683) fills in the attribute defaults
 * declared in sdl.dtd; the table `dtd_defaults` below carries those defaults so
 * no DTD file and no libxml2 are needed.  Arithmetic that ends up in the blob
 * (camera basis, transform stack, bounding boxes) repeats the reference's
 * operation order in the same precision, so the numbers are bit-identical to
 * what scene.c computes.  Build this file with -ffp-contract=off.
 */
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "front.h"
#include "xml_mini.h"

#define TWO_PI 6.283185307179586476925287   /* cgmath.h:8 */

/* ---------------------------------------------------------------- errors */

static __thread char g_error[512];

void rtamd_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof g_error, fmt, ap);
	va_end(ap);
}

const char *rtamd_last_error(void)
{
	return g_error;
}

/* ---------------------------------------------------- DTD attribute defaults */

static const struct { const char *elem, *attr, *value; } dtd_defaults[] = {
	/* sdl.dtd:6-15 */
	{ "Config", "width", "512" }, { "Config", "height", "512" },
	{ "Config", "antialiasing", "false" }, { "Config", "aa_samples", "4" },
	{ "Config", "shadow_samples", "9" }, { "Config", "reflection_samples", "10" },
	{ "Config", "max_reflections", "5" }, { "Config", "depth_of_field", "false" },
	/* sdl.dtd:26 */
	{ "Camera", "near_plane", "1.0" },
	/* sdl.dtd:36-37, 46-47 */
	{ "PointLight", "intensity", "1" }, { "PointLight", "color", "1 1 1" },
	{ "AreaLight", "intensity", "1" }, { "AreaLight", "color", "1 1 1" },
	/* sdl.dtd:77, 85 */
	{ "Cylinder", "capped", "true" }, { "Cone", "capped", "true" },
	/* sdl.dtd:111-114 */
	{ "Material", "specular_color", "1 1 1" }, { "Material", "specular_exponent", "100" },
	{ "Material", "reflect", "0" }, { "Material", "glossiness", "0" },
	/* sdl.dtd:123-124 */
	{ "Scene", "background", "0 0 0" }, { "Scene", "cubemap", "" },
};

/* xmlGetProp with DTD defaults applied; NULL when the attribute is absent and
 * has no default (#REQUIRED / #IMPLIED). */
static const char *get_prop(const XmlNode *node, const char *attr)
{
	const char *v = xml_attr(node, attr);
	if (v)
		return v;
	for (unsigned i = 0; i < sizeof dtd_defaults / sizeof dtd_defaults[0]; i++)
		if (strcmp(dtd_defaults[i].elem, node->name) == 0 &&
				strcmp(dtd_defaults[i].attr, attr) == 0)
			return dtd_defaults[i].value;
	return NULL;
}

/* Required attribute: the reference would dereference NULL here (its asserts
 * are compiled out); this front end reports it instead. */
static const char *need_prop(const XmlNode *node, const char *attr, int *bad)
{
	const char *v = get_prop(node, attr);
	if (!v) {
		rtamd_set_error("<%s> is missing required attribute \"%s\"", node->name, attr);
		*bad = 1;
		return "";
	}
	return v;
}

/* ------------------------------------------------------- value parsing */

/* scene.c:16-27: three strtof tokens widened to double. */
static void parse_vec3(const char *s, double v[3])
{
	char *end;
	v[0] = strtof(s, &end);
	v[1] = strtof(end, &end);
	v[2] = strtof(end, &end);
}

static double parse_double(const char *s)   /* scene.c:29-35 */
{
	char *end;
	return strtod(s, &end);
}

static int parse_int(const char *s)          /* scene.c:37-43 */
{
	char *end;
	return (int)strtol(s, &end, 10);
}

/* scene.c:45-56: r,g,b by strtof; the reference leaves alpha unset and never
 * outputs it (ppm.c:22-24) - it is stored as 0 here. */
static void parse_colour(const char *s, float c[4])
{
	char *end;
	c[0] = strtof(s, &end);
	c[1] = strtof(end, &end);
	c[2] = strtof(end, &end);
	c[3] = 0.0f;
}

static int parse_bool(const char *s)         /* scene.c:58-62 */
{
	return strcmp(s, "true") == 0;
}

/* ------------------------------------------------------------ small math */

static double dot3(const double a[3], const double b[3])   /* vector.c:31-34 */
{
	return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

static void scale3(double s, const double a[3], double out[3])   /* vector.c:41-49 */
{
	out[0] = a[0] * s;
	out[1] = a[1] * s;
	out[2] = a[2] * s;
}

static void normalize3(const double a[3], double out[3])   /* vector.c:51-54 */
{
	scale3(1 / sqrt(dot3(a, a)), a, out);
}

static void cross3(const double a[3], const double b[3], double c[3])   /* vector.c:56-64 */
{
	double x = a[1] * b[2] - a[2] * b[1];
	double y = a[2] * b[0] - a[0] * b[2];
	double z = a[0] * b[1] - a[1] * b[0];
	c[0] = x;
	c[1] = y;
	c[2] = z;
}

#define M4(m, i, j) (m)[4 * (j) + (i)]     /* matrix.c:66 */

static void mat4_identity(double m[16])
{
	for (int k = 0; k < 16; k++)
		m[k] = 0.0;
	M4(m, 0, 0) = M4(m, 1, 1) = M4(m, 2, 2) = M4(m, 3, 3) = 1.0;
}

static void mat4_scale(double m[16], double x, double y, double z)   /* matrix.c:190-198 */
{
	for (int k = 0; k < 16; k++)
		m[k] = 0;
	M4(m, 0, 0) = x;
	M4(m, 1, 1) = y;
	M4(m, 2, 2) = z;
	M4(m, 3, 3) = 1;
}

static void mat4_translate(double m[16], double x, double y, double z)   /* matrix.c:200-208 */
{
	mat4_identity(m);
	M4(m, 0, 3) = x;
	M4(m, 1, 3) = y;
	M4(m, 2, 3) = z;
}

/* matrix.c:215-228 via quaternion.c:88-108 and the Euler-Rodrigues form
 * quaternion.c:180-197. */
static void mat4_rotate(double m[16], double angle, double ax, double ay, double az)
{
	double qw, qx, qy, qz;
	double l2 = ax * ax + ay * ay + az * az;
	double l = sqrt(l2), s = sin(angle / 2), c = cos(angle / 2);
	if (l2 == 0.0) {
		qw = 1; qx = 0; qy = 0; qz = 0;
	} else {
		qw = c;
		qx = s * ax / l;
		qy = s * ay / l;
		qz = s * az / l;
	}
	double w = qw, x = qx, y = qy, z = qz;
	for (int k = 0; k < 16; k++)
		m[k] = 0.0;
	M4(m, 0, 0) = w * w + x * x - y * y - z * z;
	M4(m, 0, 1) = 2 * x * y - 2 * w * z;
	M4(m, 0, 2) = 2 * x * z + 2 * w * y;
	M4(m, 1, 0) = 2 * x * y + 2 * w * z;
	M4(m, 1, 1) = w * w - x * x + y * y - z * z;
	M4(m, 1, 2) = 2 * y * z - 2 * w * x;
	M4(m, 2, 0) = 2 * x * z - 2 * w * y;
	M4(m, 2, 1) = 2 * y * z + 2 * w * x;
	M4(m, 2, 2) = w * w - x * x - y * y + z * z;
	M4(m, 3, 3) = 1.0;
}

/* b * a --> a, matrix.c:138-150 */
static void mat4_lmul(const double b[16], double a[16])
{
	for (int j = 0; j < 4; j++) {
		double A0 = M4(a, 0, j), A1 = M4(a, 1, j), A2 = M4(a, 2, j), A3 = M4(a, 3, j);
		for (int i = 0; i < 4; i++)
			M4(a, i, j) = M4(b, i, 0) * A0 + M4(b, i, 1) * A1 + M4(b, i, 2) * A2 +
					M4(b, i, 3) * A3;
	}
}

/* a * b --> a, matrix.c:153-165 */
static void mat4_rmul(double a[16], const double b[16])
{
	for (int i = 0; i < 4; i++) {
		double A0 = M4(a, i, 0), A1 = M4(a, i, 1), A2 = M4(a, i, 2), A3 = M4(a, i, 3);
		for (int j = 0; j < 4; j++)
			M4(a, i, j) = A0 * M4(b, 0, j) + A1 * M4(b, 1, j) + A2 * M4(b, 2, j) +
					A3 * M4(b, 3, j);
	}
}

/* matrix.c:61-81 + vector.c:95-104: homogeneous point transform with divide. */
static void transform_point(const double m[16], const double v[3], double out[3])
{
	const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = 1.0;
	double x = M4(m, 0, 0) * v0 + M4(m, 0, 1) * v1 + M4(m, 0, 2) * v2 + M4(m, 0, 3) * v3;
	double y = M4(m, 1, 0) * v0 + M4(m, 1, 1) * v1 + M4(m, 1, 2) * v2 + M4(m, 1, 3) * v3;
	double z = M4(m, 2, 0) * v0 + M4(m, 2, 1) * v1 + M4(m, 2, 2) * v2 + M4(m, 2, 3) * v3;
	double w = M4(m, 3, 0) * v0 + M4(m, 3, 1) * v1 + M4(m, 3, 2) * v2 + M4(m, 3, 3) * v3;
	out[0] = x / w;
	out[1] = y / w;
	out[2] = z / w;
}

/* ------------------------------------------------------------ import state */

typedef struct NamedCamera {
	char *name;
	double position[3], u[3], v[3], w[3];
	float fov, near_plane;
} NamedCamera;

typedef struct NamedLight {
	char *name;
	rtamd_light light;
} NamedLight;

typedef struct NamedShape {
	char *name;
	int type;
	double edge1[3], edge2[3];
	float radius, height;
	int capped;
	int mesh;        /* index into Import.meshes */
} NamedShape;

typedef struct NamedMaterial {
	char *name;
	rtamd_material mat;
} NamedMaterial;

typedef struct CubeFace {
	int w, h;
	float *texels;   /* RGBA, bottom-up */
} CubeFace;

typedef struct Import {
	const char *asset_dir;
	rtamd_blob_header hdr;
	int have_config, have_scene;
	int ncam;  NamedCamera *cam;
	int nlight; NamedLight *light;
	int nshape; NamedShape *shape;
	int nmat;  NamedMaterial *mat;
	int nmesh; FrontMesh *meshes;
	int nscene_lights; rtamd_light scene_light[RTAMD_MAX_LIGHTS];
	int has_cubemap; CubeFace face[6];
	double cubemap_seconds;
	int nsurf, capsurf; rtamd_surface *surf;   /* in DOCUMENT order; reversed when packed */
} Import;

static __thread double t_cubemap_seconds = -1.0;

double rtamd_last_cubemap_load_seconds(void)
{
	return t_cubemap_seconds;
}

static double wall_seconds(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static char *dupstr(const char *s)
{
	size_t n = strlen(s);
	char *d = malloc(n + 1);
	memcpy(d, s, n + 1);
	return d;
}

static void asset_path(const Import *im, const char *rel, char *out, size_t outlen)
{
	if (im->asset_dir && im->asset_dir[0] && rel[0] != '/')
		snprintf(out, outlen, "%s/%s", im->asset_dir, rel);
	else
		snprintf(out, outlen, "%s", rel);
}

/* ------------------------------------------------------------ sections */

static int import_config(Import *im, const XmlNode *node)   /* scene.c:78-91 */
{
	rtamd_blob_header *h = &im->hdr;
	h->width = parse_int(get_prop(node, "width"));
	h->height = parse_int(get_prop(node, "height"));
	h->antialiasing = parse_bool(get_prop(node, "antialiasing"));
	h->aa_samples = parse_int(get_prop(node, "aa_samples"));
	h->shadow_samples = parse_int(get_prop(node, "shadow_samples"));
	h->reflection_samples = parse_int(get_prop(node, "reflection_samples"));
	h->max_reflections = parse_int(get_prop(node, "max_reflections"));
	h->depth_of_field = parse_bool(get_prop(node, "depth_of_field"));
	im->have_config = 1;
	return 0;
}

static int import_cameras(Import *im, const XmlNode *node)   /* scene.c:93-129 */
{
	int n = xml_child_count(node), i = 0, bad = 0;
	im->cam = calloc((size_t)(n ? n : 1), sizeof *im->cam);
	im->ncam = n;
	for (const XmlNode *c = node->first_child; c; c = c->next, i++) {
		NamedCamera *cam = &im->cam[i];
		double direction[3], up[3], nd[3], uxw[3];
		parse_vec3(need_prop(c, "position", &bad), cam->position);
		cam->fov = (float)parse_double(need_prop(c, "fovy", &bad));
		cam->name = dupstr(need_prop(c, "name", &bad));
		parse_vec3(need_prop(c, "direction", &bad), direction);
		parse_vec3(need_prop(c, "up", &bad), up);
		if (bad)
			return RTAMD_ERR_SCENE;
		/* scene.c:115-117 */
		normalize3(direction, nd);
		scale3(-1, nd, cam->w);
		cross3(up, cam->w, uxw);
		normalize3(uxw, cam->u);
		cross3(cam->w, cam->u, cam->v);
		cam->near_plane = (float)parse_double(get_prop(c, "near_plane"));
	}
	return 0;
}

static int import_lights(Import *im, const XmlNode *node)   /* scene.c:131-166 */
{
	int n = xml_child_count(node), i = 0, bad = 0;
	im->light = calloc((size_t)(n ? n : 1), sizeof *im->light);
	im->nlight = n;
	for (const XmlNode *c = node->first_child; c; c = c->next, i++) {
		rtamd_light *l = &im->light[i].light;
		if (strcmp(c->name, "PointLight") == 0) {
			l->type = RTAMD_LIGHT_POINT;
		} else if (strcmp(c->name, "AreaLight") == 0) {
			l->type = RTAMD_LIGHT_AREA;
			parse_vec3(need_prop(c, "edge1", &bad), l->edge1);
			parse_vec3(need_prop(c, "edge2", &bad), l->edge2);
		} else {
			rtamd_set_error("Unknown light type: \"%s\"", c->name);
			return RTAMD_ERR_SCENE;
		}
		parse_vec3(need_prop(c, "position", &bad), l->position);
		parse_colour(get_prop(c, "color"), l->colour);
		l->intensity = (float)parse_double(get_prop(c, "intensity"));
		im->light[i].name = dupstr(need_prop(c, "name", &bad));
		if (bad)
			return RTAMD_ERR_SCENE;
	}
	return 0;
}

static int import_shapes(Import *im, const XmlNode *node)   /* scene.c:168-231 */
{
	int n = xml_child_count(node), i = 0, bad = 0;
	im->shape = calloc((size_t)(n ? n : 1), sizeof *im->shape);
	im->meshes = calloc((size_t)(n ? n : 1), sizeof *im->meshes);
	im->nshape = n;
	for (const XmlNode *c = node->first_child; c; c = c->next, i++) {
		NamedShape *s = &im->shape[i];
		s->mesh = -1;
		if (strcmp(c->name, "Plane") == 0) {
			s->type = RTAMD_SHAPE_PLANE;
			parse_vec3(need_prop(c, "edge1", &bad), s->edge1);
			parse_vec3(need_prop(c, "edge2", &bad), s->edge2);
		} else if (strcmp(c->name, "Disk") == 0) {
			s->type = RTAMD_SHAPE_DISK;
			s->radius = (float)parse_double(need_prop(c, "radius", &bad));
		} else if (strcmp(c->name, "Sphere") == 0) {
			s->type = RTAMD_SHAPE_SPHERE;
			s->radius = (float)parse_double(need_prop(c, "radius", &bad));
		} else if (strcmp(c->name, "Cylinder") == 0) {
			s->type = RTAMD_SHAPE_CYLINDER;
			s->radius = (float)parse_double(need_prop(c, "radius", &bad));
			s->height = (float)parse_double(need_prop(c, "height", &bad));
			s->capped = parse_bool(get_prop(c, "capped"));
		} else if (strcmp(c->name, "Cone") == 0) {
			s->type = RTAMD_SHAPE_CONE;
			s->radius = (float)parse_double(need_prop(c, "radius", &bad));
			s->height = (float)parse_double(need_prop(c, "height", &bad));
			s->capped = parse_bool(get_prop(c, "capped"));
		} else if (strcmp(c->name, "Mesh") == 0) {
			char path[1024];
			const char *src = need_prop(c, "src", &bad);
			if (bad)
				return RTAMD_ERR_SCENE;
			s->type = RTAMD_SHAPE_MESH;
			asset_path(im, src, path, sizeof path);
			int rc = front_obj_load(path, &im->meshes[im->nmesh]);
			if (rc)
				return rc;      /* "Couldn't load mesh", scene.c:215-219 */
			s->mesh = im->nmesh++;
		} else {
			rtamd_set_error("Unknown geometry type: %s", c->name);
			return RTAMD_ERR_SCENE;
		}
		if (bad)
			return RTAMD_ERR_SCENE;
		s->name = dupstr(need_prop(c, "name", &bad));
		if (bad)
			return RTAMD_ERR_SCENE;
	}
	return 0;
}

static int import_materials(Import *im, const XmlNode *node)   /* scene.c:258-282 */
{
	int n = xml_child_count(node), i = 0, bad = 0;
	im->mat = calloc((size_t)(n ? n : 1), sizeof *im->mat);
	im->nmat = n;
	for (const XmlNode *c = node->first_child; c; c = c->next, i++) {
		rtamd_material *m = &im->mat[i].mat;
		parse_colour(need_prop(c, "diffuse_color", &bad), m->diffuse);
		parse_colour(get_prop(c, "specular_color"), m->specular);
		m->shininess = (int)parse_double(get_prop(c, "specular_exponent"));
		m->reflect = (float)parse_double(get_prop(c, "reflect"));
		m->glossiness = (float)parse_double(get_prop(c, "glossiness"));
		im->mat[i].name = dupstr(need_prop(c, "name", &bad));
		if (bad)
			return RTAMD_ERR_SCENE;
	}
	return 0;
}

/* Index of the LAST light whose name begins with the `len` characters at
 * `prefix`, or -1.  scene.c:317 compares with strncmp over the requested length
 * only, so "light" selects "light0", and an empty entry selects the last light. */
static int light_by_prefix(const Import *im, const char *prefix, size_t len)
{
	int found = -1;
	for (int j = 0; j < im->nlight; j++)
		if (strncmp(im->light[j].name, prefix, len) == 0)
			found = j;
	return found;
}

/* The `lights=` attribute of <Scene>: comma-separated entries resolved in order
 * (scene.c:284-330).  As there, the entry count is one more than the number of
 * commas and is checked before anything is resolved, and the first entry is at
 * least one character long (a leading comma is part of it). */
static int import_light_refs(Import *im, const char *names)
{
	im->nscene_lights = 0;
	if (names[0] == '\0')
		return 0;
	int entries = 1;
	for (const char *p = names; *p; p++)
		entries += *p == ',';
	if (entries > RTAMD_MAX_LIGHTS) {
		rtamd_set_error("Too many lights: %d", entries);
		return RTAMD_ERR_SCENE;
	}
	const char *entry = names;
	size_t len = 1 + strcspn(names + 1, ",");
	for (int i = 0; i < entries; i++) {
		const int j = light_by_prefix(im, entry, len);
		if (j < 0) {
			rtamd_set_error("Couldn't find light %d of %s", i, names);
			return RTAMD_ERR_SCENE;
		}
		im->scene_light[i] = im->light[j].light;
		im->nscene_lights = i + 1;
		if (entry[len] == '\0')
			break;
		entry += len + 1;
		len = strcspn(entry, ",");
	}
	im->nscene_lights = entries;
	return 0;
}

/* float min/max helpers with the reference macro's comparison direction
 * (cgmath.h:15-16) */
#define FMAX(a, b) ((a) > (b) ? (a) : (b))
#define FMIN(a, b) ((a) < (b) ? (a) : (b))

/* scene.c:458-526 (analytic shapes) and scene.c:424-456 (meshes). */
static void build_bbox(const Import *im, const NamedShape *shape, rtamd_surface *surf)
{
	float xmin = 0, ymin = 0, zmin = 0, xmax = 0, ymax = 0, zmax = 0;
	const double *m = surf->model_to_world;

	if (shape->type == RTAMD_SHAPE_MESH) {
		const FrontMesh *mesh = &im->meshes[shape->mesh];
		float b[6] = { HUGE_VALF, HUGE_VALF, HUGE_VALF, -HUGE_VALF, -HUGE_VALF, -HUGE_VALF };
		for (int i = 0; i < mesh->num_vertices; i++) {
			double pos[3];
			transform_point(m, &mesh->vertex[3 * i], pos);
			for (int a = 0; a < 3; a++) {
				if (pos[a] < b[a])
					b[a] = (float)pos[a];
				if (pos[a] > b[3 + a])
					b[3 + a] = (float)pos[a];
			}
		}
		memcpy(surf->bbox, b, sizeof b);
		return;
	}

	switch (shape->type) {
	case RTAMD_SHAPE_PLANE: {
		/* the parallelogram's corners 0, e1, e2, e1 + e2 (scene.c:465-478); the
		 * extremes are folded from the last corner to the first, the nesting of the
		 * reference's MIN / MAX macros */
		const double *e1 = shape->edge1, *e2 = shape->edge2;
		float lo[3], hi[3];
		for (int a = 0; a < 3; a++) {
			const double corner[4] = { 0.0, e1[a], e2[a], e1[a] + e2[a] };
			double mn = corner[3], mx = corner[3];
			for (int c = 2; c >= 0; c--) {
				mn = FMIN(corner[c], mn);
				mx = FMAX(corner[c], mx);
			}
			lo[a] = (float)mn;
			hi[a] = (float)mx;
		}
		xmin = lo[0]; ymin = lo[1]; zmin = lo[2];
		xmax = hi[0]; ymax = hi[1]; zmax = hi[2];
		break;
	}
	case RTAMD_SHAPE_DISK:
		xmin = ymin = -shape->radius;
		xmax = ymax = +shape->radius;
		zmin = (float)-1e-3;
		zmax = (float)+1e-3;
		break;
	case RTAMD_SHAPE_SPHERE:
		xmin = ymin = zmin = -shape->radius;
		xmax = ymax = zmax = +shape->radius;
		break;
	case RTAMD_SHAPE_CONE:
	case RTAMD_SHAPE_CYLINDER:
		xmin = ymin = -shape->radius;
		xmax = ymax = +shape->radius;
		zmin = 0;
		zmax = shape->height;
		break;
	default:
		break;
	}
	/* scene.c:517-523: float -= double constant, stored back to float */
	xmin = (float)(xmin - 1e-3);
	ymin = (float)(ymin - 1e-3);
	zmin = (float)(zmin - 1e-3);
	xmax = (float)(xmax + 1e-3);
	ymax = (float)(ymax + 1e-3);
	zmax = (float)(zmax + 1e-3);

	/* bbox.c:6-42 */
	double corner[8][3] = {
		{ xmin, ymin, zmin }, { xmax, ymin, zmin }, { xmin, ymax, zmin }, { xmax, ymax, zmin },
		{ xmin, ymin, zmax }, { xmax, ymin, zmax }, { xmin, ymax, zmax }, { xmax, ymax, zmax } };
	float b[6] = { HUGE_VALF, HUGE_VALF, HUGE_VALF, -HUGE_VALF, -HUGE_VALF, -HUGE_VALF };
	for (int i = 0; i < 8; i++) {
		double tv[3];
		transform_point(m, corner[i], tv);
		for (int a = 0; a < 3; a++) {
			if (tv[a] < b[a])
				b[a] = (float)tv[a];
			if (tv[a] > b[3 + a])
				b[3 + a] = (float)tv[a];
		}
	}
	memcpy(surf->bbox, b, sizeof b);
}

typedef struct MatLevel {
	double matrix[16], inverse[16];
} MatLevel;

/* scene.c:332-422 */
static int import_graph(Import *im, const XmlNode *node, const MatLevel *top, int depth)
{
	if (depth > 256) {
		rtamd_set_error("Scene graph nested too deeply");
		return RTAMD_ERR_LIMIT;
	}
	if (strcmp(node->name, "Shape") == 0) {
		int bad = 0;
		const char *shape_name = need_prop(node, "geometry", &bad);
		if (bad)
			return RTAMD_ERR_SCENE;
		int si = -1, mi = -1;
		for (int i = 0; i < im->nshape; i++)
			if (strcmp(im->shape[i].name, shape_name) == 0)
				si = i;
		if (si < 0) {
			rtamd_set_error("Requested shape \"%s\" not found", shape_name);
			return RTAMD_ERR_SCENE;
		}
		if (xml_attr(node, "texture")) {
			rtamd_set_error("Sorry, no texture support yet");
			return RTAMD_ERR_SCENE;
		}
		/* #IMPLIED in the DTD but required by the importer, scene.c:359-368 */
		const char *material_name = xml_attr(node, "material");
		if (!material_name) {
			rtamd_set_error("<Shape geometry=\"%s\"> has no material", shape_name);
			return RTAMD_ERR_SCENE;
		}
		for (int i = 0; i < im->nmat; i++)
			if (strcmp(im->mat[i].name, material_name) == 0)
				mi = i;
		if (mi < 0) {
			rtamd_set_error("Requested material \"%s\" not found", material_name);
			return RTAMD_ERR_SCENE;
		}
		if (im->nsurf == im->capsurf) {
			im->capsurf = im->capsurf ? 2 * im->capsurf : 16;
			im->surf = realloc(im->surf, (size_t)im->capsurf * sizeof *im->surf);
		}
		const NamedShape *shape = &im->shape[si];
		rtamd_surface *s = &im->surf[im->nsurf++];
		memset(s, 0, sizeof *s);
		s->shape_type = shape->type;
		s->material = mi;
		s->mesh = shape->mesh;
		s->capped = shape->capped;
		memcpy(s->edge1, shape->edge1, sizeof s->edge1);
		memcpy(s->edge2, shape->edge2, sizeof s->edge2);
		s->radius = shape->radius;
		s->height = shape->height;
		memcpy(s->model_to_world, top->matrix, sizeof s->model_to_world);
		memcpy(s->world_to_model, top->inverse, sizeof s->world_to_model);
		build_bbox(im, shape, s);       /* scene.c:657 */
		return 0;
	}

	double mat[16], inv[16];
	int bad = 0;
	if (strcmp(node->name, "Rotate") == 0) {
		double axis[3];
		double angle = parse_double(need_prop(node, "angle", &bad)) * TWO_PI / 360.;
		parse_vec3(need_prop(node, "axis", &bad), axis);
		mat4_rotate(mat, angle, axis[0], axis[1], axis[2]);
		mat4_rotate(inv, -angle, axis[0], axis[1], axis[2]);
	} else if (strcmp(node->name, "Translate") == 0) {
		double v[3], nv[3];
		parse_vec3(need_prop(node, "vector", &bad), v);
		mat4_translate(mat, v[0], v[1], v[2]);
		scale3(-1, v, nv);
		mat4_translate(inv, nv[0], nv[1], nv[2]);
	} else if (strcmp(node->name, "Scale") == 0) {
		double v[3];
		parse_vec3(need_prop(node, "scale", &bad), v);
		mat4_scale(mat, v[0], v[1], v[2]);
		mat4_scale(inv, 1. / v[0], 1. / v[1], 1. / v[2]);
	} else {
		rtamd_set_error("Unknown node: \"%s\"", node->name);
		return RTAMD_ERR_SCENE;
	}
	if (bad)
		return RTAMD_ERR_SCENE;

	MatLevel next = *top;               /* matstack_push copies the top, matrix.c:43-59 */
	mat4_rmul(next.matrix, mat);        /* scene.c:409 */
	mat4_lmul(inv, next.inverse);       /* scene.c:410 */
	for (const XmlNode *c = node->first_child; c; c = c->next) {
		int rc = import_graph(im, c, &next, depth + 1);
		if (rc)
			return rc;
	}
	return 0;
}

/* texture.c:16-70,118-133 + colour.c:41-74 */
static int load_cubemap(Import *im, const char *prefix)
{
	static const char *dir[6] = { "negative_x", "positive_x", "negative_y", "positive_y",
		"negative_z", "positive_z" };     /* texture.c:12-14 */
	for (int f = 0; f < 6; f++) {
		char rel[1024], path[1100];
		int w, h, ch;
		uint8_t *px;
		snprintf(rel, sizeof rel, "%s_%s.png", prefix, dir[f]);
		asset_path(im, rel, path, sizeof path);
		int rc = front_png_load(path, &w, &h, &ch, &px);
		if (rc)
			return rc;
		float *dst = malloc((size_t)w * h * 4 * sizeof(float));
		if (!dst) {
			free(px);
			return RTAMD_ERR_NOMEM;
		}
		for (int j = 0; j < h; j++)
			for (int i = 0; i < w; i++) {
				int jj = h - 1 - j;
				const uint8_t *s = px + (size_t)ch * ((size_t)j * w + i);
				float *d = dst + 4 * ((size_t)w * jj + i);
				d[0] = (float)(s[0] / 255.);
				d[1] = (float)(s[1] / 255.);
				d[2] = (float)(s[2] / 255.);
				d[3] = ch == 4 ? (float)(s[3] / 255.) : 1.0f;
			}
		free(px);
		im->face[f].w = w;
		im->face[f].h = h;
		im->face[f].texels = dst;
	}
	im->has_cubemap = 1;
	return 0;
}

static int import_scene(Import *im, const XmlNode *node)   /* scene.c:528-597 */
{
	const char *cam_name = xml_attr(node, "camera");
	if (!cam_name) {
		rtamd_set_error("At least one camera has to be defined");
		return RTAMD_ERR_SCENE;
	}
	int ci = -1;
	for (int i = 0; i < im->ncam; i++)
		if (strcmp(im->cam[i].name, cam_name) == 0)
			ci = i;
	if (ci < 0) {
		rtamd_set_error("Requested camera \"%s\" not found", cam_name);
		return RTAMD_ERR_SCENE;
	}
	const NamedCamera *cam = &im->cam[ci];
	memcpy(im->hdr.cam_position, cam->position, sizeof cam->position);
	memcpy(im->hdr.cam_u, cam->u, sizeof cam->u);
	memcpy(im->hdr.cam_v, cam->v, sizeof cam->v);
	memcpy(im->hdr.cam_w, cam->w, sizeof cam->w);
	im->hdr.cam_fov = cam->fov;
	im->hdr.cam_near_plane = cam->near_plane;

	const char *light_names = xml_attr(node, "lights");
	if (!light_names) {
		rtamd_set_error("A scene without lights is pretty dark...");
		return RTAMD_ERR_SCENE;
	}
	int rc = import_light_refs(im, light_names);
	if (rc)
		return rc;

	parse_colour(get_prop(node, "background"), im->hdr.background);
	const char *cubemap = get_prop(node, "cubemap");
	if (cubemap[0] != '\0') {
		const double t0 = wall_seconds();
		rc = load_cubemap(im, cubemap);
		im->cubemap_seconds = wall_seconds() - t0;
		if (rc)
			return rc;
	}

	/* Only the first element child is walked, scene.c:586. */
	MatLevel top;
	mat4_identity(top.matrix);
	mat4_identity(top.inverse);
	if (!node->first_child) {
		rtamd_set_error("Error importing the scene graph");
		return RTAMD_ERR_SCENE;
	}
	rc = import_graph(im, node->first_child, &top, 0);
	if (rc)
		return rc;
	im->have_scene = 1;
	return 0;
}

/* ------------------------------------------------------------ packing */

static size_t align8(size_t x)
{
	return (x + 7u) & ~(size_t)7u;
}

/* Name of the <Mesh> shape that loaded front mesh `mesh` (the first, if several do). */
static const char *mesh_shape_name(const Import *im, int mesh)
{
	for (int i = 0; i < im->nshape; i++)
		if (im->shape[i].type == RTAMD_SHAPE_MESH && im->shape[i].mesh == mesh)
			return im->shape[i].name;
	return "";
}

static int pack_blob(Import *im, void **blob_out, size_t *size_out)
{
	/* Only meshes referenced by a surface are carried; they are renumbered. */
	int *remap = calloc((size_t)(im->nmesh ? im->nmesh : 1), sizeof(int));
	int nmesh = 0;
	for (int i = 0; i < im->nmesh; i++)
		remap[i] = -1;
	for (int s = 0; s < im->nsurf; s++) {
		int m = im->surf[s].mesh;
		if (m >= 0 && remap[m] < 0)
			remap[m] = nmesh++;
	}

	size_t off = align8(sizeof(rtamd_blob_header));
	size_t off_lights = off;     off = align8(off + sizeof(rtamd_light) * (size_t)im->nscene_lights);
	size_t off_mats = off;       off = align8(off + sizeof(rtamd_material) * (size_t)im->nmat);
	size_t off_surfs = off;      off = align8(off + sizeof(rtamd_surface) * (size_t)im->nsurf);
	size_t off_meshes = off;     off = align8(off + sizeof(rtamd_mesh) * (size_t)nmesh);
	size_t off_cube = off;       off = align8(off + (im->has_cubemap ? sizeof(rtamd_cubemap) : 0));

	rtamd_mesh *mdesc = calloc((size_t)(nmesh ? nmesh : 1), sizeof *mdesc);
	for (int i = 0; i < im->nmesh; i++) {
		if (remap[i] < 0)
			continue;
		const FrontMesh *fm = &im->meshes[i];
		rtamd_mesh *d = &mdesc[remap[i]];
		d->num_vertices = fm->num_vertices;
		d->num_normals = fm->num_normals;
		d->num_triangles = fm->num_triangles;
		d->off_vertex = off;   off = align8(off + sizeof(double) * 3 * (size_t)fm->num_vertices);
		if (fm->num_normals) {
			d->off_normal = off;
			off = align8(off + sizeof(double) * 3 * (size_t)fm->num_normals);
		}
		d->off_triangle = off; off = align8(off + sizeof(int32_t) * 6 * (size_t)fm->num_triangles);
		d->off_name = off;     off = align8(off + strlen(mesh_shape_name(im, i)) + 1);
	}
	rtamd_cubemap cube;
	memset(&cube, 0, sizeof cube);
	if (im->has_cubemap)
		for (int f = 0; f < 6; f++) {
			cube.width[f] = im->face[f].w;
			cube.height[f] = im->face[f].h;
			cube.off_face[f] = off;
			off = align8(off + sizeof(float) * 4 * (size_t)im->face[f].w * im->face[f].h);
		}

	uint8_t *blob = calloc(1, off);
	if (!blob) {
		free(remap);
		free(mdesc);
		return RTAMD_ERR_NOMEM;
	}
	rtamd_blob_header *h = &im->hdr;
	h->magic = RTAMD_BLOB_MAGIC;
	h->total_size = off;
	h->num_lights = im->nscene_lights;
	h->num_materials = im->nmat;
	h->num_surfaces = im->nsurf;
	h->num_meshes = nmesh;
	h->has_cubemap = im->has_cubemap;
	h->off_lights = off_lights;
	h->off_materials = off_mats;
	h->off_surfaces = off_surfs;
	h->off_meshes = off_meshes;
	h->off_cubemap = im->has_cubemap ? off_cube : 0;
	memcpy(blob, h, sizeof *h);
	memcpy(blob + off_lights, im->scene_light, sizeof(rtamd_light) * (size_t)im->nscene_lights);
	for (int i = 0; i < im->nmat; i++)
		memcpy(blob + off_mats + sizeof(rtamd_material) * (size_t)i, &im->mat[i].mat,
				sizeof(rtamd_material));
	/* The surface list is built by prepending (scene.c:342-343): traversal
	 * order is the reverse of document order. */
	for (int i = 0; i < im->nsurf; i++) {
		rtamd_surface s = im->surf[im->nsurf - 1 - i];
		if (s.mesh >= 0)
			s.mesh = remap[s.mesh];
		memcpy(blob + off_surfs + sizeof(rtamd_surface) * (size_t)i, &s, sizeof s);
	}
	memcpy(blob + off_meshes, mdesc, sizeof(rtamd_mesh) * (size_t)nmesh);
	for (int i = 0; i < im->nmesh; i++) {
		if (remap[i] < 0)
			continue;
		const FrontMesh *fm = &im->meshes[i];
		const rtamd_mesh *d = &mdesc[remap[i]];
		memcpy(blob + d->off_vertex, fm->vertex, sizeof(double) * 3 * (size_t)fm->num_vertices);
		if (fm->num_normals)
			memcpy(blob + d->off_normal, fm->normal,
					sizeof(double) * 3 * (size_t)fm->num_normals);
		memcpy(blob + d->off_triangle, fm->triangle,
				sizeof(int32_t) * 6 * (size_t)fm->num_triangles);
		strcpy((char *)blob + d->off_name, mesh_shape_name(im, i));
	}
	if (im->has_cubemap) {
		memcpy(blob + off_cube, &cube, sizeof cube);
		for (int f = 0; f < 6; f++)
			memcpy(blob + cube.off_face[f], im->face[f].texels,
					sizeof(float) * 4 * (size_t)im->face[f].w * im->face[f].h);
	}
	free(remap);
	free(mdesc);
	*blob_out = blob;
	*size_out = off;
	return 0;
}

static void import_free(Import *im)
{
	for (int i = 0; i < im->ncam; i++)
		free(im->cam[i].name);
	for (int i = 0; i < im->nlight; i++)
		free(im->light[i].name);
	for (int i = 0; i < im->nshape; i++)
		free(im->shape[i].name);
	for (int i = 0; i < im->nmat; i++)
		free(im->mat[i].name);
	for (int i = 0; i < im->nmesh; i++)
		front_mesh_free(&im->meshes[i]);
	for (int f = 0; f < 6; f++)
		free(im->face[f].texels);
	free(im->cam);
	free(im->light);
	free(im->shape);
	free(im->mat);
	free(im->meshes);
	free(im->surf);
}

int rtamd_sdl_load(const char *sdl_path, const char *asset_dir, void **blob, size_t *blob_size)
{
	if (!sdl_path || !blob || !blob_size) {
		rtamd_set_error("rtamd_sdl_load: NULL argument");
		return RTAMD_ERR_ARG;
	}
	*blob = NULL;
	*blob_size = 0;
	FILE *f = fopen(sdl_path, "rb");
	if (!f) {
		rtamd_set_error("Error opening file %s", sdl_path);   /* scene.c:686 */
		return RTAMD_ERR_IO;
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *text = malloc((size_t)n + 1);
	if (!text) {
		fclose(f);
		return RTAMD_ERR_NOMEM;
	}
	size_t got = fread(text, 1, (size_t)n, f);
	fclose(f);
	text[got] = '\0';

	char xerr[256];
	XmlNode *root = xml_parse(text, xerr, sizeof xerr);
	free(text);
	if (!root) {
		rtamd_set_error("Error opening file %s: %s", sdl_path, xerr);
		return RTAMD_ERR_IO;
	}

	Import im;
	memset(&im, 0, sizeof im);
	im.asset_dir = asset_dir;
	im.cubemap_seconds = -1.0;
	int rc = 0;
	/* scene.c:610-652 */
	for (const XmlNode *node = root->first_child; node && !rc; node = node->next) {
		if (strcmp(node->name, "Config") == 0)
			rc = import_config(&im, node);
		else if (strcmp(node->name, "Cameras") == 0)
			rc = import_cameras(&im, node);
		else if (strcmp(node->name, "Lights") == 0)
			rc = import_lights(&im, node);
		else if (strcmp(node->name, "Geometry") == 0)
			rc = import_shapes(&im, node);
		else if (strcmp(node->name, "Textures") == 0)
			rc = 0;                                   /* scene.c:233-256: ignored */
		else if (strcmp(node->name, "Materials") == 0)
			rc = import_materials(&im, node);
		else if (strcmp(node->name, "Scene") == 0)
			rc = import_scene(&im, node);
		else {
			rtamd_set_error("Unknown node: %s", node->name);
			rc = RTAMD_ERR_SCENE;
		}
	}
	if (!rc && (!im.have_config || !im.have_scene)) {
		rtamd_set_error("Error importing SDL from file %s: missing <Config> or <Scene>",
				sdl_path);
		rc = RTAMD_ERR_SCENE;
	}
	if (!rc)
		rc = pack_blob(&im, blob, blob_size);
	t_cubemap_seconds = im.cubemap_seconds;
	import_free(&im);
	xml_free(root);
	return rc;
}

void rtamd_blob_free(void *blob)
{
	free(blob);
}
