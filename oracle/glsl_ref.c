/*
 * glsl_ref.c — TEST INFRASTRUCTURE ONLY (oracle B).
 *
 * Headless executor for the UNMODIFIED reference GLSL shaders on Mesa llvmpipe.
 * It contains no shader text: the vertex/fragment sources are read at run time
 * from a shader root (normally /root/reference/res/shaders) that the caller
 * names, compiled by Mesa's GLSL compiler and executed by llvmpipe on the host
 * CPU.  It exists to (a) generate the golden fixtures under tests/golden/
 * (tests/golden/make_golden.py) and (b) time the reference shader on CPU cores.
 * Nothing in the product path (raytracer_amd/) may link or load this file.
 *
 * How it gets a GL 4.6 context without X / EGL / OSMesa: it dlopen()s Mesa's
 * swrast_dri.so and speaks the DRI "swrast" loader interface directly
 * (GL/internal/dri_interface.h): DRI_SWRast.createNewScreen2 ->
 * createContextAttribs -> createNewDrawable -> DRI_Core.bindContext, with a
 * DRI_SWRastLoader whose put/getImage are no-ops (all rendering goes to FBOs).
 * GL entry points come from libglapi's _glapi_get_proc_address.
 *
 * The GL state it sets mirrors what the reference's Rust wrappers set:
 *   quad geometry + index order   src/gl_wrapper/geometry_set.rs:129-144
 *   SSBO upload + binding         src/gl_wrapper/buffer.rs:9-19,100-107
 *   albedo textures               src/gl_wrapper/texture.rs:11-42,65-79
 *                                 (RGBA8 internal, GL_RGB/UNSIGNED_BYTE data,
 *                                  LINEAR min/mag, wrap left at GL default)
 *   ray textures / targets        src/main.rs:60-84 (RGB32F / RGBA32F, NEAREST is
 *                                 what FramebufferManager attaches for float targets)
 *   fused-shader interface        res/shaders/ray_trace_and_shade.frag:17-21,47-51
 *   split-shader interface        res/shaders/ray_trace/ray_trace.frag:41-46
 *   ray creation                  res/shaders/ray_trace/ray_create.frag:7-9
 *
 * Build: see oracle/Makefile (output goes to oracle/_ref/, git-ignored).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

/* ------------------------------------------------------------------ */
/* GL function table                                                   */
/* ------------------------------------------------------------------ */
#define GL_FUNCS(X) \
    X(PFNGLGETSTRINGPROC, glGetString) \
    X(PFNGLGETERRORPROC, glGetError) \
    X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays) \
    X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray) \
    X(PFNGLGENBUFFERSPROC, glGenBuffers) \
    X(PFNGLDELETEBUFFERSPROC, glDeleteBuffers) \
    X(PFNGLBINDBUFFERPROC, glBindBuffer) \
    X(PFNGLBUFFERDATAPROC, glBufferData) \
    X(PFNGLBINDBUFFERBASEPROC, glBindBufferBase) \
    X(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray) \
    X(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer) \
    X(PFNGLCREATESHADERPROC, glCreateShader) \
    X(PFNGLSHADERSOURCEPROC, glShaderSource) \
    X(PFNGLCOMPILESHADERPROC, glCompileShader) \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv) \
    X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) \
    X(PFNGLDELETESHADERPROC, glDeleteShader) \
    X(PFNGLCREATEPROGRAMPROC, glCreateProgram) \
    X(PFNGLATTACHSHADERPROC, glAttachShader) \
    X(PFNGLLINKPROGRAMPROC, glLinkProgram) \
    X(PFNGLGETPROGRAMIVPROC, glGetProgramiv) \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) \
    X(PFNGLUSEPROGRAMPROC, glUseProgram) \
    X(PFNGLDELETEPROGRAMPROC, glDeleteProgram) \
    X(PFNGLUNIFORM1IPROC, glUniform1i) \
    X(PFNGLUNIFORM1FPROC, glUniform1f) \
    X(PFNGLUNIFORM1IVPROC, glUniform1iv) \
    X(PFNGLUNIFORM3FPROC, glUniform3f) \
    X(PFNGLUNIFORM4FPROC, glUniform4f) \
    X(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv) \
    X(PFNGLGENTEXTURESPROC, glGenTextures) \
    X(PFNGLDELETETEXTURESPROC, glDeleteTextures) \
    X(PFNGLBINDTEXTUREPROC, glBindTexture) \
    X(PFNGLACTIVETEXTUREPROC, glActiveTexture) \
    X(PFNGLTEXIMAGE2DPROC, glTexImage2D) \
    X(PFNGLTEXPARAMETERIPROC, glTexParameteri) \
    X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers) \
    X(PFNGLDELETEFRAMEBUFFERSPROC, glDeleteFramebuffers) \
    X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer) \
    X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D) \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) \
    X(PFNGLDRAWBUFFERSPROC, glDrawBuffers) \
    X(PFNGLVIEWPORTPROC, glViewport) \
    X(PFNGLCLEARCOLORPROC, glClearColor) \
    X(PFNGLCLEARPROC, glClear) \
    X(PFNGLDRAWELEMENTSPROC, glDrawElements) \
    X(PFNGLFINISHPROC, glFinish) \
    X(PFNGLREADPIXELSPROC, glReadPixels) \
    X(PFNGLREADBUFFERPROC, glReadBuffer) \
    X(PFNGLPIXELSTOREIPROC, glPixelStorei) \
    X(PFNGLDISABLEPROC, glDisable)

#define DECL(T, n) static T p_##n;
GL_FUNCS(DECL)
#undef DECL

static char g_err[4096];
static char g_shader_root[1024];
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static GLuint g_vao, g_vbo, g_ibo;
static int g_inited;

const char *glslref_last_error(void) { return g_err; }

static void seterr(const char *fmt, const char *a) { snprintf(g_err, sizeof g_err, fmt, a); }

/* ------------------------------------------------------------------ */
/* DRI swrast loader (no window system: all callbacks are inert)       */
/* ------------------------------------------------------------------ */
static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16;
}
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4);
}
static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 1 },
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

int glslref_init(const char *shader_root) {
    if (g_inited) { snprintf(g_shader_root, sizeof g_shader_root, "%s", shader_root); return 0; }
    /* llvmpipe natively reports GL 4.5 / GLSL 4.50; the reference shaders say
     * "#version 460 core".  The override only changes the advertised version. */
    setenv("MESA_GL_VERSION_OVERRIDE", "4.6", 1);
    setenv("MESA_GLSL_VERSION_OVERRIDE", "460", 1);
    setenv("GALLIUM_DRIVER", "llvmpipe", 1);
    snprintf(g_shader_root, sizeof g_shader_root, "%s", shader_root);

    const char *drv = getenv("GLSLREF_SWRAST");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *h = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { seterr("dlopen swrast_dri.so failed: %s", dlerror()); return -1; }
    void *hapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!hapi) { seterr("dlopen libglapi.so.0 failed: %s", dlerror()); return -1; }

    typedef const __DRIextension **(*get_ext_fn)(void);
    get_ext_fn get_ext = (get_ext_fn)dlsym(h, "__driDriverGetExtensions_swrast");
    if (!get_ext) { seterr("%s", "__driDriverGetExtensions_swrast not found"); return -1; }
    const __DRIextension **drv_exts = get_ext();
    for (int i = 0; drv_exts[i]; i++) {
        if (!strcmp(drv_exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)drv_exts[i];
        if (!strcmp(drv_exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)drv_exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) { seterr("%s", "DRI_Core / DRI_SWRast v4 missing"); return -1; }

    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, drv_exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) { seterr("%s", "createNewScreen2 failed"); return -1; }

    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { seterr("%s", "createContextAttribs failed"); return -1; }
    g_draw = g_swrast->createNewDrawable(g_screen, configs[0], NULL);
    if (!g_draw) { seterr("%s", "createNewDrawable failed"); return -1; }
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { seterr("%s", "bindContext failed"); return -1; }

    typedef void *(*gpa_fn)(const char *);
    gpa_fn gpa = (gpa_fn)dlsym(hapi, "_glapi_get_proc_address");
    if (!gpa) { seterr("%s", "_glapi_get_proc_address not found"); return -1; }
#define LOAD(T, n) p_##n = (T)gpa(#n); if (!p_##n) { seterr("GL function missing: %s", #n); return -1; }
    GL_FUNCS(LOAD)
#undef LOAD

    /* quad: geometry_set.rs:130-131 */
    static const float verts[8] = { -1, -1, -1, 1, 1, 1, 1, -1 };
    static const unsigned idx[6] = { 0, 2, 1, 0, 2, 3 };
    p_glGenVertexArrays(1, &g_vao);
    p_glBindVertexArray(g_vao);
    p_glGenBuffers(1, &g_vbo);
    p_glBindBuffer(GL_ARRAY_BUFFER, g_vbo);
    p_glBufferData(GL_ARRAY_BUFFER, sizeof verts, verts, GL_STATIC_DRAW);
    p_glEnableVertexAttribArray(0);
    p_glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 0, 0);
    p_glGenBuffers(1, &g_ibo);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, g_ibo);
    p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof idx, idx, GL_STATIC_DRAW);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    g_inited = 1;
    return 0;
}

const char *glslref_renderer(void) {
    if (!g_inited) return "";
    static char buf[512];
    snprintf(buf, sizeof buf, "%s | %s | GLSL %s", (const char *)p_glGetString(GL_RENDERER),
             (const char *)p_glGetString(GL_VERSION), (const char *)p_glGetString(GL_SHADING_LANGUAGE_VERSION));
    return buf;
}

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */
static char *read_text(const char *rel) {
    char path[2048];
    snprintf(path, sizeof path, "%s/%s", g_shader_root, rel);
    FILE *f = fopen(path, "rb");
    if (!f) { seterr("cannot open shader file %s", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *s = (char *)malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); seterr("short read on %s", path); return NULL; }
    s[n] = 0;
    fclose(f);
    return s;
}

static GLuint compile(GLenum type, const char *rel) {
    char *src = read_text(rel);
    if (!src) return 0;
    GLuint s = p_glCreateShader(type);
    const char *srcs[1] = { src };
    p_glShaderSource(s, 1, srcs, NULL);
    p_glCompileShader(s);
    free(src);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[3000]; GLsizei n = 0;
        p_glGetShaderInfoLog(s, sizeof log, &n, log);
        snprintf(g_err, sizeof g_err, "compile %s failed: %s", rel, log);
        p_glDeleteShader(s);
        return 0;
    }
    return s;
}

static GLuint build_program(const char *vert_rel, const char *frag_rel) {
    GLuint v = compile(GL_VERTEX_SHADER, vert_rel);
    if (!v) return 0;
    GLuint f = compile(GL_FRAGMENT_SHADER, frag_rel);
    if (!f) { p_glDeleteShader(v); return 0; }
    GLuint p = p_glCreateProgram();
    p_glAttachShader(p, v);
    p_glAttachShader(p, f);
    p_glLinkProgram(p);
    p_glDeleteShader(v);
    p_glDeleteShader(f);
    GLint ok = 0;
    p_glGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3000]; GLsizei n = 0;
        p_glGetProgramInfoLog(p, sizeof log, &n, log);
        snprintf(g_err, sizeof g_err, "link %s + %s failed: %s", vert_rel, frag_rel, log);
        p_glDeleteProgram(p);
        return 0;
    }
    return p;
}

static GLuint make_ssbo(GLuint binding, const void *data, size_t bytes) {
    static const uint32_t dummy[4] = { 0, 0, 0, 0 };
    GLuint b;
    p_glGenBuffers(1, &b);
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
    if (!data || bytes == 0) { data = dummy; bytes = sizeof dummy; }
    p_glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)bytes, data, GL_STATIC_DRAW);
    p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, binding, b);
    return b;
}

/* float ray texture: RGB32F, NEAREST, sampled at pixel centres. `rows_top_first`
 * arrays in this API are always given bottom row first (GL convention): the
 * Python side flips. */
static GLuint make_float_tex(int unit, GLenum ifmt, GLenum fmt, int w, int h, const float *data) {
    GLuint t;
    p_glGenTextures(1, &t);
    p_glActiveTexture(GL_TEXTURE0 + unit);
    p_glBindTexture(GL_TEXTURE_2D, t);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 4);
    p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)ifmt, w, h, 0, fmt, GL_FLOAT, data);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    return t;
}

/* albedo texture exactly as texture.rs:11-42 uploads it */
static GLuint make_albedo_tex(int unit, int w, int h, const uint8_t *rgb) {
    GLuint t;
    p_glGenTextures(1, &t);
    p_glActiveTexture(GL_TEXTURE0 + unit);
    p_glBindTexture(GL_TEXTURE_2D, t);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 4); /* GL default, never changed by the reference */
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, w, h, 0, GL_RGB, GL_UNSIGNED_BYTE, rgb);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    return t;
}

static int make_target(int w, int h, GLenum ifmt, GLuint *fbo, GLuint *tex) {
    p_glGenTextures(1, tex);
    p_glActiveTexture(GL_TEXTURE0 + 15);
    p_glBindTexture(GL_TEXTURE_2D, *tex);
    p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)ifmt, w, h, 0, ifmt == GL_RGB32F ? GL_RGB : GL_RGBA, GL_FLOAT, NULL);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glGenFramebuffers(1, fbo);
    p_glBindFramebuffer(GL_FRAMEBUFFER, *fbo);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, *tex, 0);
    GLenum bufs[1] = { GL_COLOR_ATTACHMENT0 };
    p_glDrawBuffers(1, bufs);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { seterr("%s", "framebuffer incomplete"); return -1; }
    p_glViewport(0, 0, w, h);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_BLEND);
    return 0;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int cmp_d(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

/* draw `reps` times (after `warm` untimed draws); returns the median seconds per draw */
static double draw_timed(int warm, int reps, float cr, float cg, float cb, float ca) {
    double *ts = (double *)malloc(sizeof(double) * (size_t)(reps > 0 ? reps : 1));
    for (int i = 0; i < warm + reps; i++) {
        p_glClearColor(cr, cg, cb, ca);
        p_glClear(GL_COLOR_BUFFER_BIT);
        p_glFinish();
        double t0 = now_s();
        p_glBindVertexArray(g_vao);
        p_glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, 0);
        p_glFinish();
        double t1 = now_s();
        if (i >= warm) ts[i - warm] = t1 - t0;
    }
    qsort(ts, (size_t)reps, sizeof(double), cmp_d);
    double med = reps > 0 ? ts[reps / 2] : 0.0;
    free(ts);
    return med;
}

typedef struct {
    const void *nodes; uint32_t n_nodes;          /* 36-byte records */
    const void *triangles; uint32_t n_triangles;  /* 16-byte records */
    const float *positions; uint32_t n_vertices;  /* 3 floats each */
    const float *texcoords;                       /* 2 floats each or NULL */
    const float *normals;                         /* 3 floats each or NULL */
} glslref_scene;

/* ------------------------------------------------------------------ */
/* fused shader: util/quad01.vert + ray_trace_and_shade.frag           */
/* ------------------------------------------------------------------ */
int glslref_run_fused(int w, int h, const glslref_scene *sc,
                      const float *dir_rgb, const float *org_rgb, /* w*h*3, bottom row first */
                      int has_texcoords, int has_normals,
                      const uint8_t *tex0, int tw0, int th0,
                      const uint8_t *tex1, int tw1, int th1,
                      float *out_rgba,                               /* w*h*4, bottom row first */
                      int warm, int reps, double *median_seconds) {
    if (!g_inited) { seterr("%s", "glslref_init not called"); return -1; }
    GLuint prog = build_program("util/quad01.vert", "ray_trace_and_shade.frag");
    if (!prog) return -1;
    GLuint fbo, target;
    if (make_target(w, h, GL_RGBA32F, &fbo, &target)) return -1;
    GLuint b[5];
    b[0] = make_ssbo(0, sc->nodes, (size_t)sc->n_nodes * 36);
    b[1] = make_ssbo(1, sc->triangles, (size_t)sc->n_triangles * 16);
    b[2] = make_ssbo(2, sc->positions, (size_t)sc->n_vertices * 12);
    b[3] = make_ssbo(3, sc->texcoords, sc->texcoords ? (size_t)sc->n_vertices * 8 : 0);
    b[4] = make_ssbo(4, sc->normals, sc->normals ? (size_t)sc->n_vertices * 12 : 0);
    GLuint t[4];
    t[0] = make_float_tex(0, GL_RGB32F, GL_RGB, w, h, dir_rgb);
    t[1] = make_float_tex(1, GL_RGB32F, GL_RGB, w, h, org_rgb);
    t[2] = make_albedo_tex(2, tw0, th0, tex0);
    t[3] = make_albedo_tex(3, tw1, th1, tex1);
    p_glUseProgram(prog);
    p_glUniform1i(0, 0);
    p_glUniform1i(1, 1);
    p_glUniform1i(2, has_texcoords ? 1 : 0);
    p_glUniform1i(3, has_normals ? 1 : 0);
    GLint units[2] = { 2, 3 };
    p_glUniform1iv(4, 2, units);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during fused setup", e); return -1; }
    double med = draw_timed(warm, reps < 1 ? 1 : reps, 0, 0, 0, 0);
    if (median_seconds) *median_seconds = med;
    p_glReadBuffer(GL_COLOR_ATTACHMENT0);
    p_glReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, out_rgba);
    e = p_glGetError();
    p_glUseProgram(0);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glDeleteProgram(prog);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(1, &target);
    p_glDeleteTextures(4, t);
    p_glDeleteBuffers(5, b);
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during fused draw/readback", e); return -1; }
    return 0;
}

/* ------------------------------------------------------------------ */
/* split shader: util/quad01.vert + ray_trace/ray_trace.frag           */
/* target cleared to (1e30,0,0,0) exactly as main.rs:179 does          */
/* ------------------------------------------------------------------ */
int glslref_run_trace(int w, int h, const glslref_scene *sc,
                      const float *dir_rgb, const float *org_rgb,
                      float *out_rgba, int warm, int reps, double *median_seconds) {
    if (!g_inited) { seterr("%s", "glslref_init not called"); return -1; }
    GLuint prog = build_program("util/quad01.vert", "ray_trace/ray_trace.frag");
    if (!prog) return -1;
    GLuint fbo, target;
    if (make_target(w, h, GL_RGBA32F, &fbo, &target)) return -1;
    GLuint b[3];
    b[0] = make_ssbo(0, sc->nodes, (size_t)sc->n_nodes * 36);
    b[1] = make_ssbo(1, sc->triangles, (size_t)sc->n_triangles * 16);
    b[2] = make_ssbo(2, sc->positions, (size_t)sc->n_vertices * 12);
    GLuint t[2];
    t[0] = make_float_tex(0, GL_RGB32F, GL_RGB, w, h, dir_rgb);
    t[1] = make_float_tex(1, GL_RGB32F, GL_RGB, w, h, org_rgb);
    p_glUseProgram(prog);
    p_glUniform1i(0, 0);
    p_glUniform1i(1, 1);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during trace setup", e); return -1; }
    double med = draw_timed(warm, reps < 1 ? 1 : reps, 1e30f, 0, 0, 0);
    if (median_seconds) *median_seconds = med;
    p_glReadBuffer(GL_COLOR_ATTACHMENT0);
    p_glReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, out_rgba);
    e = p_glGetError();
    p_glUseProgram(0);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glDeleteProgram(prog);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(1, &target);
    p_glDeleteTextures(2, t);
    p_glDeleteBuffers(3, b);
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during trace draw/readback", e); return -1; }
    return 0;
}

/* ------------------------------------------------------------------ */
/* ray creation: util/quad-11.vert + ray_trace/ray_create.frag         */
/* inv_proj_view is column-major, uploaded untransposed (shader.rs:160) */
/* ------------------------------------------------------------------ */
int glslref_run_raycreate(int w, int h, const float *inv_proj_view, float near_, float far_, float *out_rgb) {
    if (!g_inited) { seterr("%s", "glslref_init not called"); return -1; }
    GLuint prog = build_program("util/quad-11.vert", "ray_trace/ray_create.frag");
    if (!prog) return -1;
    GLuint fbo, target;
    if (make_target(w, h, GL_RGB32F, &fbo, &target)) return -1;
    p_glUseProgram(prog);
    p_glUniformMatrix4fv(0, 1, GL_FALSE, inv_proj_view);
    p_glUniform1f(1, near_);
    p_glUniform1f(2, far_);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during raycreate setup", e); return -1; }
    draw_timed(0, 1, 0, 0, 0, 0);
    p_glReadBuffer(GL_COLOR_ATTACHMENT0);
    p_glReadPixels(0, 0, w, h, GL_RGB, GL_FLOAT, out_rgb);
    e = p_glGetError();
    p_glUseProgram(0);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glDeleteProgram(prog);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(1, &target);
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during raycreate draw/readback", e); return -1; }
    return 0;
}

/* ------------------------------------------------------------------ */
/* secondary-ray generation: util/quad01.vert + ray_trace/ray_dispatcher.frag */
/* G-buffer textures NEAREST (framebuffer_manager.rs:33), blue noise    */
/* uploaded like every image texture (texture.rs:11-42, LINEAR), four   */
/* RGB32F colour attachments cleared to 0 = NO_RAY (main.rs:68-72,156-158) */
/* ------------------------------------------------------------------ */
int glslref_run_dispatch(int w, int h, const float *position_rgb, const float *normal_mat_rgba,
                         const uint8_t *noise_rgb, int nw, int nh,
                         const float *light_pos, const float *camera_pos, const float *noise_offset_scale,
                         float *out_org, float *out_shadow, float *out_reflect, float *out_ambient) {
    if (!g_inited) { seterr("%s", "glslref_init not called"); return -1; }
    GLuint prog = build_program("util/quad01.vert", "ray_trace/ray_dispatcher.frag");
    if (!prog) return -1;
    GLuint fbo, targets[4];
    p_glGenFramebuffers(1, &fbo);
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glGenTextures(4, targets);
    GLenum bufs[4];
    for (int k = 0; k < 4; k++) {
        p_glActiveTexture(GL_TEXTURE0 + 15);
        p_glBindTexture(GL_TEXTURE_2D, targets[k]);
        p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB32F, w, h, 0, GL_RGB, GL_FLOAT, NULL);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + k, GL_TEXTURE_2D, targets[k], 0);
        bufs[k] = GL_COLOR_ATTACHMENT0 + k;
    }
    p_glDrawBuffers(4, bufs);
    if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { seterr("%s", "dispatch framebuffer incomplete"); return -1; }
    p_glViewport(0, 0, w, h);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_BLEND);
    GLuint mats = make_ssbo(0, NULL, 0);   /* Material { bool reflect; }: read but never used (:36) */
    GLuint t[3];
    t[0] = make_float_tex(0, GL_RGB32F, GL_RGB, w, h, position_rgb);
    t[1] = make_float_tex(1, GL_RGBA32F, GL_RGBA, w, h, normal_mat_rgba);
    t[2] = make_albedo_tex(2, nw, nh, noise_rgb);
    p_glUseProgram(prog);
    p_glUniform1i(0, 0);
    p_glUniform1i(1, 1);
    p_glUniform1i(2, 2);
    p_glUniform3f(3, light_pos[0], light_pos[1], light_pos[2]);
    p_glUniform3f(4, camera_pos[0], camera_pos[1], camera_pos[2]);
    p_glUniform4f(5, noise_offset_scale[0], noise_offset_scale[1], noise_offset_scale[2], noise_offset_scale[3]);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during dispatch setup", e); return -1; }
    draw_timed(0, 1, 0, 0, 0, 0);
    float *outs[4] = { out_org, out_shadow, out_reflect, out_ambient };
    for (int k = 0; k < 4; k++) {
        p_glReadBuffer(GL_COLOR_ATTACHMENT0 + k);
        p_glReadPixels(0, 0, w, h, GL_RGB, GL_FLOAT, outs[k]);
    }
    e = p_glGetError();
    p_glUseProgram(0);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    p_glDeleteProgram(prog);
    p_glDeleteFramebuffers(1, &fbo);
    p_glDeleteTextures(4, targets);
    p_glDeleteTextures(3, t);
    p_glDeleteBuffers(1, &mats);
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x during dispatch draw/readback", e); return -1; }
    return 0;
}
