/*
 * ref_harness.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Driver that runs the REFERENCE's own hot path (ray.c, shading.c, lighting.c,
 * texture.c, mesh.c kd-tree, math) on a scene blob.  It is compiled by
 * oracle/Makefile together with the reference sources where they lie under
 * /root/reference; none of them is copied here.  The reference's front end
 * (scene.c + libxml2, objreader) is not used: objreader's parser is generated
 * by flex/bison, which this image does not have, so the Scene/Config globals
 * the hot path reads (scene.h:105-106) are filled from the blob instead.
 *
 * Only available where /root/reference exists; outputs go to files that the
 * tests under tests/ compare against and that tests/golden/ keeps small
 * samples of.
 *
 * Usage:
 *   ref_harness render <blob> <out_prefix> [--seed S] [--hits] [--rays]
 *                      [--window x0 y0 x1 y1] [--no-ppm]
 *   ref_harness trace  <blob> <rays.f64> <out.hits>
 *
 * render repeats raytracer.c:55-99 (srand, row-major pixel loop, pixel_colour
 * at raytracer.c:32-54) call for call.  With --window only pixels inside the
 * window are shaded; pixels before them in raster order are skipped by
 * discarding the 2*aa^2 draws their camera rays would have made, which is
 * exact only for scenes where nothing else draws (no area light, no glossy
 * material).
 */
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "colour.h"
#include "ray.h"
#include "shading.h"
#include "ppm.h"

#include "rtamd.h"

static Config the_config;
static Scene the_scene;
static Camera the_camera;
static Surface **surface_by_ordinal;
static int num_surfaces;

static void *xcalloc(size_t n, size_t sz)
{
	void *p = calloc(n ? n : 1, sz);
	if (!p) {
		fprintf(stderr, "out of memory\n");
		exit(2);
	}
	return p;
}

static Vec3 v3(const double *d)
{
	return (Vec3) { d[0], d[1], d[2] };
}

static unsigned char *read_file(const char *path, size_t *len)
{
	FILE *f = fopen(path, "rb");
	if (!f) {
		fprintf(stderr, "cannot open %s\n", path);
		exit(2);
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	unsigned char *buf = xcalloc((size_t)n + 1, 1);
	if (fread(buf, 1, (size_t)n, f) != (size_t)n) {
		fprintf(stderr, "short read on %s\n", path);
		exit(2);
	}
	fclose(f);
	*len = (size_t)n;
	return buf;
}

static void build_scene(const unsigned char *blob)
{
	const rtamd_blob_header *h = (const rtamd_blob_header *)blob;
	if (h->magic != RTAMD_BLOB_MAGIC) {
		fprintf(stderr, "not a scene blob\n");
		exit(2);
	}
	the_config.width = h->width;
	the_config.height = h->height;
	the_config.antialiasing = h->antialiasing != 0;
	the_config.aa_samples = h->aa_samples;
	the_config.shadow_samples = h->shadow_samples;
	the_config.reflection_samples = h->reflection_samples;
	the_config.max_reflections = h->max_reflections;
	the_config.depth_of_field = h->depth_of_field != 0;
	config = &the_config;

	the_camera.position = v3(h->cam_position);
	the_camera.u = v3(h->cam_u);
	the_camera.v = v3(h->cam_v);
	the_camera.w = v3(h->cam_w);
	the_camera.fov = h->cam_fov;
	the_camera.near_plane = h->cam_near_plane;
	the_scene.camera = &the_camera;

	const rtamd_light *bl = (const rtamd_light *)(blob + h->off_lights);
	Light *lights = xcalloc((size_t)h->num_lights, sizeof(Light));
	the_scene.num_lights = h->num_lights;
	for (int i = 0; i < h->num_lights; i++) {
		lights[i].type = bl[i].type == RTAMD_LIGHT_AREA ? LIGHT_AREA : LIGHT_POINT;
		lights[i].position = v3(bl[i].position);
		lights[i].plane.edge1 = v3(bl[i].edge1);
		lights[i].plane.edge2 = v3(bl[i].edge2);
		lights[i].colour = (Colour) { bl[i].colour[0], bl[i].colour[1], bl[i].colour[2],
			bl[i].colour[3] };
		lights[i].intensity = bl[i].intensity;
		the_scene.light[i] = &lights[i];
	}
	the_scene.background = (Colour) { h->background[0], h->background[1], h->background[2],
		h->background[3] };

	const rtamd_material *bm = (const rtamd_material *)(blob + h->off_materials);
	Material *mats = xcalloc((size_t)h->num_materials, sizeof(Material));
	for (int i = 0; i < h->num_materials; i++) {
		mats[i].diffuse_colour = (Colour) { bm[i].diffuse[0], bm[i].diffuse[1],
			bm[i].diffuse[2], bm[i].diffuse[3] };
		mats[i].specular_colour = (Colour) { bm[i].specular[0], bm[i].specular[1],
			bm[i].specular[2], bm[i].specular[3] };
		mats[i].shininess = bm[i].shininess;
		mats[i].reflect = bm[i].reflect;
		mats[i].glossiness = bm[i].glossiness;
	}

	const rtamd_mesh *bmesh = (const rtamd_mesh *)(blob + h->off_meshes);
	Mesh *meshes = xcalloc((size_t)h->num_meshes, sizeof(Mesh));
	for (int m = 0; m < h->num_meshes; m++) {
		Mesh *mesh = &meshes[m];
		const double *bv = (const double *)(blob + bmesh[m].off_vertex);
		const double *bn = (const double *)(blob + bmesh[m].off_normal);
		const int32_t *bt = (const int32_t *)(blob + bmesh[m].off_triangle);
		mesh->num_vertices = bmesh[m].num_vertices;
		mesh->num_normals = bmesh[m].num_normals;
		mesh->num_triangles = bmesh[m].num_triangles;
		mesh->has_normals = mesh->num_normals > 0;
		mesh->vertex = xcalloc((size_t)mesh->num_vertices, sizeof(Vec3));
		mesh->normal = xcalloc((size_t)mesh->num_normals, sizeof(Vec3));
		mesh->triangle = xcalloc((size_t)mesh->num_triangles, sizeof(Triangle));
		for (int i = 0; i < mesh->num_vertices; i++)
			mesh->vertex[i] = v3(bv + 3 * i);
		for (int i = 0; i < mesh->num_normals; i++)
			mesh->normal[i] = v3(bn + 3 * i);
		for (int i = 0; i < mesh->num_triangles; i++)
			for (int c = 0; c < 3; c++) {
				mesh->triangle[i].vertex_index[c] = bt[6 * i + c];
				mesh->triangle[i].normal_index[c] = bt[6 * i + 3 + c];
				mesh->triangle[i].texcoord_index[c] = -1;
			}
		mesh->kd_tree = NULL;
		mesh_build_kd_tree(mesh);              /* scene.c:664 */
	}

	const rtamd_surface *bs = (const rtamd_surface *)(blob + h->off_surfaces);
	num_surfaces = h->num_surfaces;
	surface_by_ordinal = xcalloc((size_t)num_surfaces, sizeof(Surface *));
	Shape *shapes = xcalloc((size_t)num_surfaces, sizeof(Shape));
	Surface *prev = NULL;
	the_scene.root = NULL;
	for (int i = 0; i < num_surfaces; i++) {
		Surface *s = xcalloc(1, sizeof(Surface));
		Shape *sh = &shapes[i];
		switch (bs[i].shape_type) {
		case RTAMD_SHAPE_PLANE:
			sh->type = SHAPE_PLANE;
			sh->u.plane.edge1 = v3(bs[i].edge1);
			sh->u.plane.edge2 = v3(bs[i].edge2);
			break;
		case RTAMD_SHAPE_DISK:
			sh->type = SHAPE_DISK;
			sh->u.disk.radius = bs[i].radius;
			break;
		case RTAMD_SHAPE_SPHERE:
			sh->type = SHAPE_SPHERE;
			sh->u.sphere.radius = bs[i].radius;
			break;
		case RTAMD_SHAPE_CYLINDER:
			sh->type = SHAPE_CYLINDER;
			sh->u.cylinder.radius = bs[i].radius;
			sh->u.cylinder.height = bs[i].height;
			sh->u.cylinder.capped = bs[i].capped != 0;
			break;
		case RTAMD_SHAPE_CONE:
			sh->type = SHAPE_CONE;
			sh->u.cone.radius = bs[i].radius;
			sh->u.cone.height = bs[i].height;
			sh->u.cone.capped = bs[i].capped != 0;
			break;
		default:
			sh->type = SHAPE_MESH;
			sh->u.mesh = &meshes[bs[i].mesh];
			break;
		}
		s->shape = sh;
		s->material = &mats[bs[i].material];
		memcpy(s->model_to_world, bs[i].model_to_world, sizeof(Mat4));
		memcpy(s->world_to_model, bs[i].world_to_model, sizeof(Mat4));
		s->bbox = (BBox) { bs[i].bbox[0], bs[i].bbox[1], bs[i].bbox[2], bs[i].bbox[3],
			bs[i].bbox[4], bs[i].bbox[5] };
		s->next = NULL;
		if (prev)
			prev->next = s;
		else
			the_scene.root = s;
		prev = s;
		surface_by_ordinal[i] = s;
	}

	the_scene.environment_map = NULL;
	if (h->has_cubemap) {
		const rtamd_cubemap *bc = (const rtamd_cubemap *)(blob + h->off_cubemap);
		CubeMap *map = xcalloc(1, sizeof(CubeMap));
		for (int f = 0; f < 6; f++) {
			Texture *t = xcalloc(1, sizeof(Texture));
			t->width = bc->width[f];
			t->height = bc->height[f];
			t->buffer = xcalloc((size_t)t->width * t->height, sizeof(Colour));
			memcpy(t->buffer, blob + bc->off_face[f],
					(size_t)t->width * t->height * sizeof(Colour));
			map->texture[f] = t;
		}
		the_scene.environment_map = map;
	}
	scene = &the_scene;
}

static void fill_record(rtamd_hit_record *rec, bool did_hit, const Hit *hit)
{
	memset(rec, 0, sizeof *rec);
	rec->surface = -1;
	rec->triangle = -1;
	if (!did_hit)
		return;
	for (int s = 0; s < num_surfaces; s++)
		if (surface_by_ordinal[s] == hit->surface)
			rec->surface = s;
	rec->t = (float)hit->t;
	rec->position[0] = hit->position.x;
	rec->position[1] = hit->position.y;
	rec->position[2] = hit->position.z;
	rec->normal[0] = hit->normal.x;
	rec->normal[1] = hit->normal.y;
	rec->normal[2] = hit->normal.z;
}

static void write_file(const char *prefix, const char *suffix, const void *data, size_t len)
{
	char path[2048];
	snprintf(path, sizeof path, "%s%s", prefix, suffix);
	FILE *f = fopen(path, "wb");
	if (!f || fwrite(data, 1, len, f) != len) {
		fprintf(stderr, "cannot write %s\n", path);
		exit(2);
	}
	fclose(f);
}

static int cmd_render(int argc, char **argv)
{
	size_t len;
	unsigned seed = 0x20071208;            /* raytracer.c:74 */
	bool want_hits = false, want_rays = false, want_ppm = true;
	int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
	const unsigned char *blob = read_file(argv[2], &len);
	const char *prefix = argv[3];
	for (int a = 4; a < argc; a++) {
		if (!strcmp(argv[a], "--seed") && a + 1 < argc)
			seed = (unsigned)strtoul(argv[++a], NULL, 0);
		else if (!strcmp(argv[a], "--hits"))
			want_hits = true;
		else if (!strcmp(argv[a], "--rays"))
			want_rays = true;
		else if (!strcmp(argv[a], "--no-ppm"))
			want_ppm = false;
		else if (!strcmp(argv[a], "--window") && a + 4 < argc) {
			x0 = atoi(argv[++a]);
			y0 = atoi(argv[++a]);
			x1 = atoi(argv[++a]);
			y1 = atoi(argv[++a]);
		} else {
			fprintf(stderr, "unknown option %s\n", argv[a]);
			return 2;
		}
	}
	build_scene(blob);
	const int width = config->width, height = config->height;
	const int spp = config->antialiasing ? config->aa_samples * config->aa_samples : 1;
	const bool windowed = x1 >= 0;
	if (!windowed) {
		x1 = width;
		y1 = height;
	}
	Colour *buffer = xcalloc((size_t)width * height, sizeof(Colour));
	size_t nsamples = (size_t)(x1 - x0) * (size_t)(y1 - y0) * (size_t)spp;
	rtamd_hit_record *hits = want_hits ? xcalloc(nsamples, sizeof *hits) : NULL;
	double *rays = want_rays ? xcalloc(nsamples * 8, sizeof(double)) : NULL;
	Camera *cam = scene->camera;
	size_t sidx = 0;

	struct timespec ts0, ts1;
	clock_gettime(CLOCK_MONOTONIC, &ts0);
	srand(seed);
	for (int j = 0; j < height; j++)
		for (int i = 0; i < width; i++) {
			if (i < x0 || i >= x1 || j < y0 || j >= y1) {
				if (windowed && config->antialiasing)
					for (int d = 0; d < 2 * spp; d++)
						(void)rand();
				continue;
			}
			Colour c;
			Ray r;
			Hit hit;
			/* raytracer.c:38-51 */
			if (config->antialiasing) {
				c = BLACK;
				for (int k = 0; k < spp; k++) {
					r = camera_ray_aa(cam, i, j, k, cam->near_plane);
					if (hits)
						fill_record(&hits[sidx], ray_intersect(r, &hit), &hit);
					if (rays) {
						double *o = rays + 8 * sidx;
						o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
						o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
						o[6] = r.near; o[7] = r.far;
					}
					sidx++;
					c = colour_add(c, ray_colour(r, 0));
				}
				c = colour_scale(1.0 / (spp), c);
			} else {
				r = camera_ray(cam, i, j, 1);
				if (hits)
					fill_record(&hits[sidx], ray_intersect(r, &hit), &hit);
				if (rays) {
					double *o = rays + 8 * sidx;
					o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
					o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
					o[6] = r.near; o[7] = r.far;
				}
				sidx++;
				c = ray_colour(r, 0);
			}
			buffer[width * j + i] = c;
		}
	clock_gettime(CLOCK_MONOTONIC, &ts1);
	/* The next value of the stream tells a test how many draws were made. */
	printf("next_rand %d\n", rand());
	printf("render_seconds %.6f\n", (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec));

	write_file(prefix, ".colour.f32", buffer, (size_t)width * height * sizeof(Colour));
	if (hits)
		write_file(prefix, ".hits.bin", hits, nsamples * sizeof *hits);
	if (rays)
		write_file(prefix, ".rays.f64", rays, nsamples * 8 * sizeof(double));
	if (want_ppm) {
		char path[2048];
		snprintf(path, sizeof path, "%s.ppm", prefix);
		FILE *out = fopen(path, "w");
		if (!out || !ppm_write(buffer, width, height, out)) {   /* raytracer.c:93-94 */
			fprintf(stderr, "cannot write %s\n", path);
			return 2;
		}
		fclose(out);
	}
	return 0;
}

static int cmd_trace(char **argv)
{
	size_t len, rlen;
	const unsigned char *blob = read_file(argv[2], &len);
	const double *rays = (const double *)read_file(argv[3], &rlen);
	size_t n = rlen / (8 * sizeof(double));
	build_scene(blob);
	rtamd_hit_record *hits = xcalloc(n, sizeof *hits);
	for (size_t i = 0; i < n; i++) {
		const double *o = rays + 8 * i;
		Ray r;
		Hit hit;
		r.origin = v3(o);
		r.direction = v3(o + 3);
		r.near = (float)o[6];
		r.far = (float)o[7];
		fill_record(&hits[i], ray_intersect(r, &hit), &hit);
	}
	write_file(argv[4], "", hits, n * sizeof *hits);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc >= 4 && !strcmp(argv[1], "render"))
		return cmd_render(argc, argv);
	if (argc == 5 && !strcmp(argv[1], "trace"))
		return cmd_trace(argv);
	fprintf(stderr, "usage: ref_harness render <blob> <prefix> [--seed S] [--hits] [--rays] "
			"[--window x0 y0 x1 y1] [--no-ppm]\n"
			"       ref_harness trace <blob> <rays.f64> <out.hits>\n");
	return 2;
}
