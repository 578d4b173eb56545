// host_check.cc - runs the C++ host classes (one C-ABI call per reference seam) and the
// device-resident tracker (svo_track_frame) on the same PGM sequence and compares their poses
// frame by frame.  usage: host_check <sequence_dir> <n_frames>   (frames: image_0/NNNNNN.pgm ...;
// optional detection boxes in <sequence_dir>/boxes/<k+1>.txt, 4 ints per line: left right top bottom)
#include <cmath>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "Tracking.h"
#include "png_reader.h"

using namespace svo_host;

int main(int argc, char** argv) {
  if (argc != 3 && argc != 4) { std::cerr << "usage: host_check <sequence_dir> <n_frames> [dense|msa]" << std::endl; return 2; }
  const std::string seq = argv[1];
  const int n = atoi(argv[2]);
  // "dense": ELAS map as the depth source on both sides; "msa": MSA map (the reference's live configuration)
  const int depth_source = argc == 4 ? (std::string(argv[3]) == "msa" ? 2 : 1) : 0;
  const svo_camera cam{718.856f, 718.856f, 607.1928f, 185.2157f, 386.1448f};
  Tracking* host = nullptr;
  svo_ctx* dev = nullptr;
  std::ofstream f(seq + "/host_kitti.txt"); f << std::fixed;
  std::ofstream f2(seq + "/host_tum.txt"); f2 << std::fixed;
  double worst = 0;
  for (int k = 0; k < n; ++k) {
    std::stringstream a, b;
    a << seq << "/image_0/" << std::setfill('0') << std::setw(6) << k << ".pgm";
    b << seq << "/image_1/" << std::setfill('0') << std::setw(6) << k << ".pgm";
    GrayImage L, R;
    if (!read_pgm(a.str(), L) || !read_pgm(b.str(), R)) { std::cerr << "cannot read " << a.str() << std::endl; return 2; }
    if (!host) {
      host = new Tracking(cam, L.cols, L.rows, 0);
      host->depth_source = depth_source;
      if (svo_create(&dev, 0, L.cols, L.rows, 500, 1) != SVO_OK || svo_track_reset(dev, &cam) != SVO_OK) return 3;
      if (svo_set_option(dev, "depth_source", depth_source) != SVO_OK) return 3;
    }
    if (k == 0) {   // frame::ElasMatch: dense disparity through the same context
      frame probe;
      probe.ctx = dev;
      const int valid = probe.ElasMatch(L, R);
      std::cout << "elas_valid " << valid << " of " << (size_t)L.cols * L.rows << std::endl;
    }
    std::vector<std::vector<int>> boxes;
    std::vector<int32_t> flat;
    {
      std::stringstream bp; bp << seq << "/boxes/" << (k + 1) << ".txt";
      std::ifstream bfile(bp.str());
      int l, r, t, b2;
      while (bfile >> l >> r >> t >> b2) { boxes.push_back({l, r, t, b2}); flat.insert(flat.end(), {l, r, t, b2}); }
    }
    host->Track(L, R, 0.1 * k, f, f2, boxes);
    svo_track_result res;
    if (svo_track_frame(dev, L.ptr(), L.cols, R.ptr(), R.cols, 0.1 * k, flat.empty() ? nullptr : flat.data(),
                        (int)boxes.size(), &res) != SVO_OK) return 4;
    if (getenv("SVO_HOST_DEBUG") && k > 0) {
      svo_pnp_stats ps; double Tp[16];
      if (svo_debug_track_pnp(dev, &ps, Tp) == SVO_OK)
        fprintf(stderr, "dev  pnp: n %d best %d inliers %d iters %d t %.9g %.9g %.9g\n", ps.n_points, ps.best_hypothesis, ps.n_inliers, ps.iterations, Tp[3], Tp[7], Tp[11]);
    }
    double d = 0;
    for (int i = 0; i < 16; ++i) d = std::max(d, (double)std::fabs(res.Tcw[i] - host->lastframe.Tcw.m[i]));
    worst = std::max(worst, d);
    std::cout << "frame " << k << " edges " << res.n_lm_edges << " local_map(host) " << host->LocalMapPoints.size()
              << " local_map(dev) " << res.n_local_map << " max|dTcw| " << d << std::endl;
    if ((int)host->LocalMapPoints.size() != res.n_local_map) { std::cout << "LOCAL MAP MISMATCH" << std::endl; return 5; }
  }
  std::cout << "worst " << worst << std::endl;
  svo_destroy(dev);
  delete host;
  return worst < 1e-4 ? 0 : 6;
}
