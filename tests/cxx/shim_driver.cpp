// Minimal C++ caller of include/ctr_shim.hpp written the way run_io_reprojection_test.cpp:189-223 drives the
// reference's classes. Reads raw f32 images + the binary point/cam file, writes 6 x f64. Used by the tests.
//   shim_driver imgA.f32 imgB.f32 w h infile outfile lv_f lv_l psz maxiter normdp_ratio donorm dopatchnorm maxpttrack
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ctr_shim.hpp"

using namespace CTR;

static std::vector<float> read_f32(const char *fn, size_t n) {
  std::vector<float> v(n);
  FILE *f = fopen(fn, "rb");
  if (!f || fread(v.data(), sizeof(float), n, f) != n) { fprintf(stderr, "cannot read %s\n", fn); exit(2); }
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc != 15) { fprintf(stderr, "usage: see source\n"); return 2; }
  const int w = atoi(argv[3]), h = atoi(argv[4]);
  optparam op;
  ictr_optparam_init(&op, atoi(argv[7]), atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), (float)atof(argv[11]),
                     atoi(argv[12]), atoi(argv[13]), atoi(argv[14]), 0);
  std::vector<float> ia = read_f32(argv[1], (size_t)w * h), ib = read_f32(argv[2], (size_t)w * h);
  // ReadPointCamFile (run_io_reprojection_test.cpp:54-79)
  FILE *f = fopen(argv[5], "rb");
  if (!f) return 2;
  double cpos_p[6], cpos_p_out[6];
  float fc[2], cc[2];
  uint32_t whu[2];
  uint64_t n;
  if (fread(cpos_p, 8, 6, f) != 6 || fread(fc, 4, 2, f) != 2 || fread(cc, 4, 2, f) != 2 || fread(whu, 4, 2, f) != 2 ||
      fread(&n, 8, 1, f) != 1)
    return 2;
  std::vector<double> pt3d(3 * n);
  if (fread(pt3d.data(), 8, 3 * n, f) != 3 * n) return 2;
  fclose(f);
  int wh[2] = {(int)whu[0], (int)whu[1]};
  try {
    Pyramid *pa = util_constructpyramide(ia.data(), w, h, op.lv_f, true, op.psz);
    Pyramid *pb = util_constructpyramide(ib.data(), w, h, op.lv_f, true, op.psz);
    const CamClass camobj(op.lv_f + 1, fc, cc, wh, op.psz);
    PoseClass posobj(&camobj, &op);
    OdometerClass odomobj(&posobj, &op);
    odomobj.Set3Dpoints(pt3d.data(), (int)n);
    odomobj.SetPose(cpos_p, *pa, *pb);
    odomobj.TrackPose(cpos_p_out);
    delete pa;
    delete pb;
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  f = fopen(argv[6], "wb");
  fwrite(cpos_p_out, 8, 6, f);  // WritePoseResult (run_io_reprojection_test.cpp:83-97)
  fclose(f);
  return 0;
}
