// ref_elas_wrap.cpp - thin C wrapper (this repo's own code) around the REAL libelas of the
// reference (Thirdparty/libelas/src/*.cpp, compiled where it lies by oracle/Makefile.ref).
// TEST INFRASTRUCTURE ONLY: it is the compiled-reference oracle for SURVEY.md section 8 row f-2
// (dense ELAS stereo on the GPU).  libelas is vendored by the reference but never called by its
// own code (include/frame.h:15 includes the header only).
//
//  ref_elas_process : Elas::process untouched (elas.cpp:32-150).
//  ref_elas_staged  : the same stage sequence, called one stage at a time through the class's
//                     private members (opened with the usual test-only `#define private public`;
//                     no reference source is modified or copied), so that every intermediate
//                     (descriptors, support points, triangles + planes, grids, the disparity maps
//                     after each post-processing step) can be tapped, and so that the triangle
//                     lists can be overridden: Triangle's enumeration order is an artefact of its
//                     memory pool, the product emits a canonical order, and the stages downstream
//                     of the triangulation are compared on identical triangle lists.
//  Uninitialised reads: libelas reads malloc'ed-but-never-written memory (descriptor rows/columns
//  2 and N-3, elas.cpp:700 with descriptor.cpp:86; D_tmp in adaptiveMean, elas.cpp:1299).  glibc's
//  M_PERTURB = 0xff makes malloc hand out zero-filled blocks (fresh mmap'ed blocks are zero anyway),
//  which makes the reference deterministic ("uninitialised = 0"); the product defines it the same way.
#include <malloc.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <iostream>
#include <vector>

#define private public
#include "elas.h"
#undef private
#include "descriptor.h"

extern "C" {

struct ref_elas_params {  // same fields, same order as Elas::parameters (elas.h:60-83), bools as int32
  int32_t disp_min, disp_max;
  float support_threshold;
  int32_t support_texture, candidate_stepsize, incon_window_size, incon_threshold, incon_min_support;
  int32_t add_corners, grid_size;
  float beta, gamma, sigma, sradius;
  int32_t match_texture, lr_threshold;
  float speckle_sim_threshold;
  int32_t speckle_size, ipol_gap_width, filter_median, filter_adaptive_mean, postprocess_only_left;
  int32_t subsampling;
};

struct ref_elas_taps {
  uint8_t *desc1, *desc2;              // W*H*16 each
  int32_t* support;                    // (u,v,d) triples
  int32_t n_support, cap_support;
  int32_t *tri1, *tri2;                // (c1,c2,c3) triples as used downstream
  float *planes1, *planes2;            // (t1a,t1b,t1c,t2a,t2b,t2c) per triangle
  int32_t n_tri1, n_tri2, cap_tri;
  int32_t *grid1, *grid2;              // (disp_max+2)*grid_w*grid_h
  float *D1_raw, *D2_raw, *D1_lr, *D2_lr, *D1_seg, *D2_seg, *D1_gap, *D2_gap, *D1_mean, *D2_mean;
  const int32_t *tri1_in, *tri2_in;    // optional override of the triangle lists
  int32_t n_tri1_in, n_tri2_in;
};

static void pin_malloc() {
  static bool done = false;
  if (!done) { mallopt(M_PERTURB, 0xff); done = true; }
}

static Elas::parameters to_ref(const ref_elas_params* p) {
  Elas::parameters q;
  q.disp_min = p->disp_min; q.disp_max = p->disp_max; q.support_threshold = p->support_threshold;
  q.support_texture = p->support_texture; q.candidate_stepsize = p->candidate_stepsize;
  q.incon_window_size = p->incon_window_size; q.incon_threshold = p->incon_threshold;
  q.incon_min_support = p->incon_min_support; q.add_corners = p->add_corners != 0;
  q.grid_size = p->grid_size; q.beta = p->beta; q.gamma = p->gamma; q.sigma = p->sigma;
  q.sradius = p->sradius; q.match_texture = p->match_texture; q.lr_threshold = p->lr_threshold;
  q.speckle_sim_threshold = p->speckle_sim_threshold; q.speckle_size = p->speckle_size;
  q.ipol_gap_width = p->ipol_gap_width; q.filter_median = p->filter_median != 0;
  q.filter_adaptive_mean = p->filter_adaptive_mean != 0;
  q.postprocess_only_left = p->postprocess_only_left != 0; q.subsampling = p->subsampling != 0;
  return q;
}

void ref_elas_default_params(int32_t middlebury, ref_elas_params* p) {
  Elas::parameters q(middlebury ? Elas::MIDDLEBURY : Elas::ROBOTICS);
  p->disp_min = q.disp_min; p->disp_max = q.disp_max; p->support_threshold = q.support_threshold;
  p->support_texture = q.support_texture; p->candidate_stepsize = q.candidate_stepsize;
  p->incon_window_size = q.incon_window_size; p->incon_threshold = q.incon_threshold;
  p->incon_min_support = q.incon_min_support; p->add_corners = q.add_corners;
  p->grid_size = q.grid_size; p->beta = q.beta; p->gamma = q.gamma; p->sigma = q.sigma;
  p->sradius = q.sradius; p->match_texture = q.match_texture; p->lr_threshold = q.lr_threshold;
  p->speckle_sim_threshold = q.speckle_sim_threshold; p->speckle_size = q.speckle_size;
  p->ipol_gap_width = q.ipol_gap_width; p->filter_median = q.filter_median;
  p->filter_adaptive_mean = q.filter_adaptive_mean; p->postprocess_only_left = q.postprocess_only_left;
  p->subsampling = q.subsampling;
}

int ref_elas_process(const uint8_t* left, const uint8_t* right, int width, int height,
                     int bytes_per_line, const ref_elas_params* p, float* D1, float* D2) {
  pin_malloc();
  Elas elas(to_ref(p));
  const int32_t dims[3] = {width, height, bytes_per_line};
  elas.process(const_cast<uint8_t*>(left), const_cast<uint8_t*>(right), D1, D2, dims);
  return 0;
}

static void snap(float* dst, const float* src, size_t n) { if (dst) memcpy(dst, src, n * sizeof(float)); }

// Stage sequence of Elas::process (elas.cpp:32-150) with taps; returns <0 on too few support points.
int ref_elas_staged(const uint8_t* left, const uint8_t* right, int width, int height, int pitch,
                    const ref_elas_params* p, ref_elas_taps* t, float* D1, float* D2) {
  pin_malloc();
  Elas e(to_ref(p));
  const Elas::parameters& param = e.param;
  e.width = width; e.height = height; e.bpl = width + 15 - (width - 1) % 16;
  const size_t n = (size_t)width * height;
  const size_t nd = p->subsampling ? (size_t)(width / 2) * (height / 2) : n;   // disparity map size
  e.I1 = (uint8_t*)_mm_malloc((size_t)e.bpl * height, 16);
  e.I2 = (uint8_t*)_mm_malloc((size_t)e.bpl * height, 16);
  memset(e.I1, 0, (size_t)e.bpl * height); memset(e.I2, 0, (size_t)e.bpl * height);
  for (int v = 0; v < height; ++v) {
    memcpy(e.I1 + (size_t)v * e.bpl, left + (size_t)v * pitch, width);
    memcpy(e.I2 + (size_t)v * e.bpl, right + (size_t)v * pitch, width);
  }
  int rc = 0;
  {
    Descriptor desc1(e.I1, width, height, e.bpl, param.subsampling);
    Descriptor desc2(e.I2, width, height, e.bpl, param.subsampling);
    if (t->desc1) memcpy(t->desc1, desc1.I_desc, n * 16);
    if (t->desc2) memcpy(t->desc2, desc2.I_desc, n * 16);
    std::vector<Elas::support_pt> sp = e.computeSupportMatches(desc1.I_desc, desc2.I_desc);
    t->n_support = (int32_t)sp.size();
    if (t->support)
      for (int i = 0; i < std::min(t->n_support, t->cap_support); ++i) {
        t->support[3 * i] = sp[i].u; t->support[3 * i + 1] = sp[i].v; t->support[3 * i + 2] = sp[i].d;
      }
    if (sp.size() < 3) { rc = -1; }
    else {
      std::vector<Elas::triangle> tri1, tri2;
      if (t->tri1_in) for (int i = 0; i < t->n_tri1_in; ++i)
        tri1.push_back(Elas::triangle(t->tri1_in[3 * i], t->tri1_in[3 * i + 1], t->tri1_in[3 * i + 2]));
      else tri1 = e.computeDelaunayTriangulation(sp, 0);
      if (t->tri2_in) for (int i = 0; i < t->n_tri2_in; ++i)
        tri2.push_back(Elas::triangle(t->tri2_in[3 * i], t->tri2_in[3 * i + 1], t->tri2_in[3 * i + 2]));
      else tri2 = e.computeDelaunayTriangulation(sp, 1);
      e.computeDisparityPlanes(sp, tri1, 0);
      e.computeDisparityPlanes(sp, tri2, 1);
      t->n_tri1 = (int32_t)tri1.size(); t->n_tri2 = (int32_t)tri2.size();
      for (int s = 0; s < 2; ++s) {
        const std::vector<Elas::triangle>& tr = s ? tri2 : tri1;
        int32_t* ti = s ? t->tri2 : t->tri1; float* pl = s ? t->planes2 : t->planes1;
        for (int i = 0; i < std::min((int32_t)tr.size(), t->cap_tri); ++i) {
          if (ti) { ti[3 * i] = tr[i].c1; ti[3 * i + 1] = tr[i].c2; ti[3 * i + 2] = tr[i].c3; }
          if (pl) { pl[6 * i] = tr[i].t1a; pl[6 * i + 1] = tr[i].t1b; pl[6 * i + 2] = tr[i].t1c;
                    pl[6 * i + 3] = tr[i].t2a; pl[6 * i + 4] = tr[i].t2b; pl[6 * i + 5] = tr[i].t2c; }
        }
      }
      const int32_t gw = (int32_t)ceil((float)width / (float)param.grid_size);
      const int32_t gh = (int32_t)ceil((float)height / (float)param.grid_size);
      int32_t grid_dims[3] = {param.disp_max + 2, gw, gh};
      const size_t gn = (size_t)(param.disp_max + 2) * gw * gh;
      int32_t* g1 = (int32_t*)calloc(gn, sizeof(int32_t));
      int32_t* g2 = (int32_t*)calloc(gn, sizeof(int32_t));
      e.createGrid(sp, g1, grid_dims, 0);
      e.createGrid(sp, g2, grid_dims, 1);
      if (t->grid1) memcpy(t->grid1, g1, gn * sizeof(int32_t));
      if (t->grid2) memcpy(t->grid2, g2, gn * sizeof(int32_t));
      e.computeDisparity(sp, tri1, g1, grid_dims, desc1.I_desc, desc2.I_desc, 0, D1);
      e.computeDisparity(sp, tri2, g2, grid_dims, desc1.I_desc, desc2.I_desc, 1, D2);
      snap(t->D1_raw, D1, nd); snap(t->D2_raw, D2, nd);
      e.leftRightConsistencyCheck(D1, D2);
      snap(t->D1_lr, D1, nd); snap(t->D2_lr, D2, nd);
      e.removeSmallSegments(D1);
      if (!param.postprocess_only_left) e.removeSmallSegments(D2);
      snap(t->D1_seg, D1, nd); snap(t->D2_seg, D2, nd);
      e.gapInterpolation(D1);
      if (!param.postprocess_only_left) e.gapInterpolation(D2);
      snap(t->D1_gap, D1, nd); snap(t->D2_gap, D2, nd);
      if (param.filter_adaptive_mean) {
        e.adaptiveMean(D1);
        if (!param.postprocess_only_left) e.adaptiveMean(D2);
      }
      snap(t->D1_mean, D1, nd); snap(t->D2_mean, D2, nd);
      if (param.filter_median) {
        e.median(D1);
        if (!param.postprocess_only_left) e.median(D2);
      }
      free(g1); free(g2);
    }
  }
  _mm_free(e.I1); _mm_free(e.I2);
  return rc;
}

// Elas::computeDelaunayTriangulation (elas.cpp:445-503) on bare points: support points (x, y, d = 0).
int ref_elas_delaunay(const int32_t* xy, int32_t n, int32_t* tri, int32_t cap) {
  pin_malloc();
  Elas::parameters param;
  Elas e(param);
  std::vector<Elas::support_pt> sp;
  for (int i = 0; i < n; ++i) sp.push_back(Elas::support_pt(xy[2 * i], xy[2 * i + 1], 0));
  std::vector<Elas::triangle> t = e.computeDelaunayTriangulation(sp, 0);
  for (int i = 0; i < std::min((int32_t)t.size(), cap); ++i) {
    tri[3 * i] = t[i].c1; tri[3 * i + 1] = t[i].c2; tri[3 * i + 2] = t[i].c3;
  }
  return (int)t.size();
}

}  // extern "C"
