// Native caller of include/ctr_shim.hpp restating run_track_nposes.cpp:133-361 (the reference's production driver:
// hundreds of pose hypotheses, forward / backward tracking chains, patch-NCC verification) with the reference's own
// control flow: ONE OdometerClass reused for every sample, Set3Dpoints / SetPose / Get2DPoints / TrackPose per chain
// link, op.dopatchnorm flipped through the aliased optparam at :281. Same input text file, same output text file
// (ReadInputFile :39-103, WriteResult :106-131); images are binary PGM (P5) instead of cv::imread.
//   nposes_driver infile outfile [--check]
// --check: also restates the NCC lines :271-355 literally on util_getPatch (host arithmetic in double) and fails when
// it differs from the device score by more than 2e-5.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "ctr_shim.hpp"

using namespace CTR;
using std::string;
using std::vector;

static bool read_pgm(const string &fn, vector<float> &img, int &w, int &h) {
  FILE *f = fopen(fn.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  int vals[3], got = 0;
  if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P5") != 0) { fclose(f); return false; }
  while (got < 3) {
    int c = fgetc(f);
    if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
    if (c == EOF) { fclose(f); return false; }
    if (isspace(c)) continue;
    ungetc(c, f);
    if (fscanf(f, "%d", &vals[got]) != 1) { fclose(f); return false; }
    ++got;
  }
  fgetc(f);  // the single whitespace after maxval
  w = vals[0];
  h = vals[1];
  if (vals[2] > 255) { fclose(f); return false; }
  vector<unsigned char> raw((size_t)w * h);
  const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
  fclose(f);
  img.assign(raw.begin(), raw.end());  // convertTo(CV_32F)
  return ok;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: nposes_driver infile outfile [--check]\n"); return 2; }
  const bool check_ncc = argc > 3 && strcmp(argv[3], "--check") == 0;
  optparam op;
  float fc[2], cc[2];
  int wh[2], fbframes[2], nocorresp = 0, nosamples = 0;
  vector<string> filenames;
  vector<vector<double>> pt3d, pt2d, poses;
  vector<vector<int>> inlids;
  {  // ReadInputFile (run_track_nposes.cpp:39-103)
    std::ifstream infile(argv[1]);
    if (!infile) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    string line;
    int lv_f, lv_l, psz, maxiter, donorm, dopatchnorm, maxpttrack, verbosity;
    float ratio;
    getline(infile, line);
    std::stringstream ls(line);
    ls >> lv_f >> lv_l >> psz >> maxiter >> ratio >> donorm >> dopatchnorm >> maxpttrack >> verbosity;
    ictr_optparam_init(&op, lv_f, lv_l, psz, maxiter, ratio, donorm, dopatchnorm, maxpttrack, verbosity);  // pads to x4 (:48-51)
    getline(infile, line); ls.str(line); ls.clear();
    ls >> fc[0] >> fc[1] >> cc[0] >> cc[1] >> wh[0] >> wh[1];
    getline(infile, line); ls.str(line); ls.clear();
    ls >> fbframes[0] >> fbframes[1];
    for (int i = 0; i < fbframes[0] + fbframes[1] + 1; ++i) {
      getline(infile, line); ls.str(line); ls.clear();
      string t;
      ls >> t;
      filenames.push_back(t);
    }
    getline(infile, line); ls.str(line); ls.clear();
    ls >> nocorresp;
    pt3d.assign(nocorresp, vector<double>(3));
    pt2d.assign(nocorresp, vector<double>(2));
    for (int i = 0; i < nocorresp; ++i) {
      getline(infile, line); ls.str(line); ls.clear();
      ls >> pt2d[i][0] >> pt2d[i][1] >> pt3d[i][0] >> pt3d[i][1] >> pt3d[i][2];
    }
    getline(infile, line); ls.str(line); ls.clear();
    ls >> nosamples;
    poses.assign(nosamples, vector<double>(6));
    inlids.resize(nosamples);
    for (int i = 0; i < nosamples; ++i) {
      getline(infile, line); ls.str(line); ls.clear();
      for (int j = 0; j < 6; ++j) ls >> poses[i][j];
      int noids;
      ls >> noids;
      inlids[i].resize(noids);
      for (int j = 0; j < noids; ++j) ls >> inlids[i][j];
    }
  }
  const int noimages = (int)filenames.size();
  try {
    vector<Pyramid *> img_ao_pyr(noimages);
    for (int i = 0; i < noimages; ++i) {  // :160-181
      vector<float> im;
      int w, h;
      if (!read_pgm(filenames[i], im, w, h)) { fprintf(stderr, "cannot read %s\n", filenames[i].c_str()); return 2; }
      img_ao_pyr[i] = util_constructpyramide(im.data(), w, h, op.lv_f, true, op.psz);
    }
    const CamClass camobj(op.lv_f + 1, fc, cc, wh, op.psz);
    PoseClass posobj(&camobj, &op);
    OdometerClass odomobj(&posobj, &op);

    vector<vector<double>> out_corr(nosamples);
    vector<vector<vector<double>>> out_pose(nosamples);
    double worst = 0.0;
    for (int sid = 0; sid < nosamples; ++sid) {  // :193
      const int nopoints = (int)inlids[sid].size();
      out_corr[sid].resize(nopoints);
      out_pose[sid].assign(noimages, vector<double>(6));
      vector<double> pt3d_in(3 * (size_t)nopoints);
      vector<float> mids(6 * (size_t)nopoints);  // x_back y_back x_ref y_ref x_fwd y_fwd
      for (int i = 0; i < nopoints; ++i) {
        const int ptid = inlids[sid][i] - 1;
        pt3d_in[i] = pt3d[ptid][0];
        pt3d_in[i + nopoints] = pt3d[ptid][1];
        pt3d_in[i + 2 * nopoints] = pt3d[ptid][2];
      }
      odomobj.Set3Dpoints(pt3d_in.data(), nopoints);
      auto reproject = [&](const double *p, float *x, float *y) {  // :219-226, 241-247, 260-266 (images are dummies)
        odomobj.SetPose(p, *img_ao_pyr[0], *img_ao_pyr[0]);
        const float *tt = odomobj.Get2DPoints();
        for (int i = 0; i < nopoints; ++i) {
          x[i] = tt[i];
          y[i] = tt[i + op.maxpttrack];
        }
      };
      reproject(&poses[sid][0], &mids[2 * nopoints], &mids[3 * nopoints]);
      double cpos_p[6];
      memcpy(cpos_p, &poses[sid][0], sizeof(double) * 6);
      memcpy(&out_pose[sid][fbframes[0]][0], cpos_p, sizeof(double) * 6);
      for (int fr = 0; fr < fbframes[1]; ++fr) {  // forward track :229-239
        const int fr_t = fr + fbframes[0];
        odomobj.SetPose(cpos_p, *img_ao_pyr[fr_t], *img_ao_pyr[fr_t + 1]);
        odomobj.TrackPose(cpos_p);
        memcpy(&out_pose[sid][fr_t + 1][0], cpos_p, sizeof(double) * 6);
      }
      reproject(cpos_p, &mids[4 * nopoints], &mids[5 * nopoints]);
      memcpy(cpos_p, &poses[sid][0], sizeof(double) * 6);
      for (int fr = 0; fr < fbframes[0]; ++fr) {  // backward track :249-258
        const int fr_t = fbframes[0] - fr;
        odomobj.SetPose(cpos_p, *img_ao_pyr[fr_t], *img_ao_pyr[fr_t - 1]);
        odomobj.TrackPose(cpos_p);
        memcpy(&out_pose[sid][fr_t - 1][0], cpos_p, sizeof(double) * 6);
      }
      reproject(cpos_p, &mids[0], &mids[nopoints]);

      op.dopatchnorm = true;  // :281 -- and it stays on: every later sample is TRACKED with patch normalisation too
      vector<float> corr(nopoints);
      util_patchNCC(*img_ao_pyr[0], *img_ao_pyr[fbframes[0]], *img_ao_pyr[noimages - 1], op.lv_l, mids.data(), nopoints, &op,
                    (float)(fbframes[0] * fbframes[0]), (float)(fbframes[1] * fbframes[1]), corr.data());
      for (int i = 0; i < nopoints; ++i) out_corr[sid][i] = corr[i];

      if (check_ncc) {  // the same lines literally, on util_getPatch
        const int n = op.psz * op.psz;
        vector<float> pb(n), pr(n), pf(n);
        const float swo = camobj.getswo(op.lv_l), sho = camobj.getsho(op.lv_l);
        for (int i = 0; i < nopoints; ++i) {
          bool val[3];
          const Pyramid *src[3] = {img_ao_pyr[0], img_ao_pyr[fbframes[0]], img_ao_pyr[noimages - 1]};
          float *dst[3] = {pb.data(), pr.data(), pf.data()};
          for (int k = 0; k < 3; ++k) {
            const float mid[2] = {mids[(2 * k) * nopoints + i], mids[(2 * k + 1) * nopoints + i]};
            val[k] = (mid[0] > 0) & (mid[1] > 0) & (mid[0] < swo) & (mid[1] < sho);
            if (val[k]) util_getPatch(*src[k], op.lv_l, mid, dst[k], &op);
          }
          double c = -1;
          if (val[1]) {
            auto unit = [&](vector<float> &p) {
              double s = 0;
              for (float v : p) s += (double)v * v;
              s = std::sqrt(s);
              for (float &v : p) v = (float)(v / s);
            };
            auto dot = [&](const vector<float> &a, const vector<float> &b) {
              double s = 0;
              for (int q = 0; q < n; ++q) s += (double)a[q] * b[q];
              return s;
            };
            if (val[0]) unit(pb);
            unit(pr);
            if (val[2]) unit(pf);
            const double w0 = val[0] ? fbframes[0] * fbframes[0] : 0, w1 = val[2] ? fbframes[1] * fbframes[1] : 0;
            const double cbr = val[0] ? std::max(0.0, dot(pb, pr)) : -1, crf = val[2] ? std::max(0.0, dot(pr, pf)) : -1;
            c = (cbr * w0 + crf * w1) / (w0 + w1);
            c = std::isnan(c) ? 0.0 : std::max(0.0, c);  // std::max(0.0f, NaN) = 0
          }
          worst = std::max(worst, std::fabs(c - (double)corr[i]));
        }
      }
    }
    if (check_ncc) {
      fprintf(stderr, "nposes_driver: max |NCC restated on util_getPatch - util_patchNCC| = %.3g\n", worst);
      if (!(worst <= 2e-5)) return 3;
    }
    {  // WriteResult (run_track_nposes.cpp:106-131)
      std::ofstream outfile(argv[2], std::ofstream::out);
      for (int sid = 0; sid < nosamples; ++sid) {
        outfile << std::setprecision(8);
        for (int j = 0; j < noimages; ++j) {
          for (int k = 0; k < 6; ++k) outfile << out_pose[sid][j][k] << " ";
          outfile << std::endl;
        }
        outfile << std::setprecision(3);
        for (size_t j = 0; j < out_corr[sid].size(); ++j) outfile << out_corr[sid][j] << " ";
        outfile << std::endl;
      }
    }
    for (Pyramid *p : img_ao_pyr) delete p;
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
