// The tracker construction of the reference's UpdaterCamera constructor (REF: PL-VIWO/src/update/cam/UpdaterCamera.cpp:26-45,
// the monocular branch) with the two class names changed — TrackKLT -> TrackKLT_HIP, TrackLSD -> TrackLSD_HIP — and nothing else:
// same arguments, same containers.  tests/test_host_shims.py compiles it (syntax only) against the stand-ins of tests/host_stub/.
#include <deque>
#include <map>
#include <memory>

#include "TrackKLT_HIP.h"
#include "TrackLSD_HIP.h"
#include "feat/FeatureDatabase.h"
#include "state/State.h"

using namespace std;
using namespace ov_core;
using namespace viw;

struct UpdaterCameraCallSite {
  shared_ptr<State> state;
  map<int, deque<double>> t_hist;
  map<int, shared_ptr<FeatureDatabase>> trackDATABASE;
  map<int, shared_ptr<TrackBase>> trackFEATS;
  map<int, shared_ptr<TrackLSD_HIP>> trackLSDS;
  shared_ptr<FeatureDatabase> point_used;

  explicit UpdaterCameraCallSite(shared_ptr<State> state) : state(state) {
    shared_ptr<OptionsCamera> op = state->op->cam;
    for (int i = 0; i < op->max_n; i++) {
      if (trackDATABASE.find(i) == trackDATABASE.end()) {
        t_hist.insert({i, deque<double>()});
        trackDATABASE.insert({i, make_shared<FeatureDatabase>()});
        trackFEATS.insert(
            {i, shared_ptr<TrackBase>(new TrackKLT_HIP(state->cam_intrinsic_model, op->n_pts, 0, op->use_stereo, op->histogram, op->fast, op->grid_x, op->grid_y, op->min_px_dist))});
        if (op->use_lines) {
          point_used = make_shared<FeatureDatabase>();
          trackLSDS.insert({i, shared_ptr<TrackLSD_HIP>(new TrackLSD_HIP(state->cam_intrinsic_model, op->use_stereo, op->histogram, trackFEATS))});
        }
      }
    }
  }

  // UpdaterCamera::feed_measurement's two tracker calls (UpdaterCamera.cpp:105-109)
  void feed(const CameraData &message, std::vector<Eigen::Vector2d> &vanishing_points) {
    int cam_id = message.sensor_ids.at(0);
    trackFEATS.at(cam_id)->feed_new_camera(message);
    if (!(trackLSDS.find(cam_id) == trackLSDS.end())) {
      trackLSDS.at(cam_id)->feed_new_camera(message, vanishing_points);
    }
  }
};
