// stereo_kitti.cc - the reference's driver (main.cpp:100-210) over the MI355X front end.
// usage: stereo_kitti <vocabulary (ignored, as in the reference)> <settings.yaml> <sequence_dir>
// Sequence layout as main.cpp:20-57: times.txt, image_2/NNNNNN.png, image_3/NNNNNN.png (gray
// image_0/image_1 and .pgm are accepted too).  Writes cameratrajectory_kitti.txt / _tum.txt and
// prints the median / mean tracking time exactly like main.cpp:200-208.  No GUI, no pacing sleep,
// offline detection boxes: <sequence_dir>/boxes/<ni+1>.txt (optional; 4 ints per line,
// left right top bottom - main.cpp:82-95).
// stereo_kitti --pipelined <vocabulary> <settings.yaml> <sequence_dir> [frames per call, default 32]: the same sequence through
// Tracking::TrackBatch (svo_track_batch_host): this thread decodes the next frames while uploads, front end and ordered tail of
// the earlier ones run; same trajectory files; reports frames per second over the whole loop (decoding included).
#include <algorithm>
#include <chrono>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "Tracking.h"
#include "convert.h"
#include "png_reader.h"

using namespace svo_host;

static bool exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }

int main(int argc, char** argv) {
  if (argc == 4 && std::string(argv[1]) == "--decode") {   // image codec self-test: png/pgm -> pgm
    GrayImage img;
    if (!read_image(argv[2], img)) return 1;
    FILE* o = fopen(argv[3], "wb");
    if (!o) return 1;
    fprintf(o, "P5\n%d %d\n255\n", img.cols, img.rows);
    fwrite(img.data.data(), 1, img.data.size(), o);
    fclose(o);
    return 0;
  }
  if (argc == 11 && std::string(argv[1]) == "--quat") {   // convert::toQuaternion self-test
    Mat33f R;
    for (int i = 0; i < 9; ++i) R.m[i] = (float)atof(argv[2 + i]);
    const std::vector<float> q = convert::toQuaternion(R);
    std::cout << std::fixed << std::setprecision(7) << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << std::endl;
    return 0;
  }
  bool pipelined = false;
  int per_call = 32;
  if (argc >= 5 && std::string(argv[1]) == "--pipelined") {
    pipelined = true;
    if (argc == 6) per_call = std::max(1, std::min(256, atoi(argv[5])));
    for (int i = 1; i < 4; ++i) argv[i] = argv[i + 1];
    argc = 4;
  }
  if (argc != 4) {
    std::cerr << "Usage: ./stereo_kitti [--pipelined] path_to_vocabulary path_to_settings path_to_sequence [frames_per_call]" << std::endl;
    return 1;
  }
  const std::string seq = argv[3];
  std::vector<double> vTimestamps;
  {
    std::ifstream fTimes(seq + "/times.txt");
    std::string s;
    while (std::getline(fTimes, s))
      if (!s.empty()) vTimestamps.push_back(atof(s.c_str()));
  }
  const int nImages = (int)vTimestamps.size();
  if (nImages == 0) { std::cerr << "no times.txt in " << seq << std::endl; return 1; }
  auto name = [&](const char* dir, int i, const char* ext) {
    std::stringstream ss;
    ss << seq << "/" << dir << "/" << std::setfill('0') << std::setw(6) << i << ext;
    return ss.str();
  };
  const char* dl = "image_2"; const char* dr = "image_3"; const char* ext = ".png";
  if (!exists(name(dl, 0, ext))) { dl = "image_0"; dr = "image_1"; }
  if (!exists(name(dl, 0, ext))) { ext = ".pgm"; }
  if (!exists(name(dl, 0, ext))) { dl = "image_2"; dr = "image_3"; }
  Tracking* mpTracker = new Tracking(argv[2]);
  std::ofstream f("cameratrajectory_kitti.txt"); f << std::fixed;
  std::ofstream f2("cameratrajectory_tum.txt"); f2 << std::fixed;
  std::vector<float> vTimesTrack(nImages);
  std::cout << std::endl << "-------" << std::endl << "Start processing sequence ..." << std::endl
            << "Images in the sequence: " << nImages << std::endl << std::endl;
  auto read_boxes = [&](int ni) {
    std::vector<std::vector<int>> boxes;
    std::stringstream bp; bp << seq << "/boxes/" << (ni + 1) << ".txt";
    std::ifstream bf(bp.str());
    int l, r, t, b;
    while (bf >> l >> r >> t >> b) boxes.push_back({l, r, t, b});
    return boxes;
  };
  if (pipelined) {
    mpTracker->batch_capacity = per_call;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> bufL, bufR;
    for (int n0 = 0; n0 < nImages; n0 += per_call) {
      const int n = std::min(per_call, nImages - n0);
      std::vector<std::vector<std::vector<int>>> boxes((size_t)n);
      size_t fb = 0;
      int cols = 0;
      for (int k = 0; k < n; ++k) {
        GrayImage imLeft, imRight;
        if (!read_image(name(dl, n0 + k, ext), imLeft) || !read_image(name(dr, n0 + k, ext), imRight)) {
          std::cerr << std::endl << "Failed to load image at: " << name(dl, n0 + k, ext) << std::endl;
          return 1;
        }
        if (k == 0) { fb = imLeft.data.size(); cols = imLeft.cols; bufL.resize(fb * n); bufR.resize(fb * n); }
        if (imLeft.data.size() != fb || imRight.data.size() != fb) { std::cerr << "image size changes within the sequence" << std::endl; return 1; }
        memcpy(bufL.data() + fb * k, imLeft.data.data(), fb);
        memcpy(bufR.data() + fb * k, imRight.data.data(), fb);
        boxes[k] = read_boxes(n0 + k);
      }
      // (pageable buffers: the call returns when they are staged, so they are refilled at once while the GPU works)
      mpTracker->TrackBatch(bufL.data(), bufR.data(), cols, n, &vTimestamps[n0], boxes);
    }
    mpTracker->FinishBatches(f, f2);
    const double total = std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();
    f.close(); f2.close();
    std::cout << std::endl << "trajectory saved!" << std::endl << "-------" << std::endl << std::endl;
    std::cout << "pipelined: " << nImages << " frames, " << per_call << " per call" << std::endl;
    std::cout << "mean tracking time: " << total / nImages << std::endl;
    std::cout << "frames per second: " << nImages / total << std::endl;
    delete mpTracker;
    return 0;
  }
  for (int ni = 0; ni < nImages; ++ni) {
    GrayImage imLeft, imRight;
    if (!read_image(name(dl, ni, ext), imLeft) || !read_image(name(dr, ni, ext), imRight)) {
      std::cerr << std::endl << "Failed to load image at: " << name(dl, ni, ext) << std::endl;
      return 1;
    }
    std::vector<std::vector<int>> boxes;
    {
      std::stringstream bp; bp << seq << "/boxes/" << (ni + 1) << ".txt";
      std::ifstream bf(bp.str());
      int l, r, t, b;
      while (bf >> l >> r >> t >> b) boxes.push_back({l, r, t, b});
    }
    const auto t1 = std::chrono::steady_clock::now();
    mpTracker->Track(imLeft, imRight, vTimestamps[ni], f, f2, boxes);
    const auto t2 = std::chrono::steady_clock::now();
    vTimesTrack[ni] = (float)std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
  }
  f.close(); f2.close();
  std::cout << std::endl << "trajectory saved!" << std::endl;
  std::sort(vTimesTrack.begin(), vTimesTrack.end());
  float totaltime = 0;
  for (int ni = 0; ni < nImages; ++ni) totaltime += vTimesTrack[ni];
  std::cout << "-------" << std::endl << std::endl;
  std::cout << "median tracking time: " << vTimesTrack[nImages / 2] << std::endl;
  std::cout << "mean tracking time: " << totaltime / nImages << std::endl;
  delete mpTracker;
  return 0;
}
