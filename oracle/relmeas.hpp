// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of the relative inter-robot measurement -> pose-index matching
//   sloam::FindRelativeMeasurementMatch  backend/sloam/src/core/sloam.cpp:321-412
//   sloam::GetIndexClosestPoseMstPair    backend/sloam/src/core/sloam.cpp:428-440
// Pinned by the reference's live test src/test/sloam_test.cpp:20-205 (cases transcribed as
// data in tests/test_oracle_pins.py).  ros::Time is restated as (sec, nsec) with
// ros::Duration's normalisation (nsec in [0, 1e9)) and toSec() = sec + 1e-9 * nsec.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <vector>

namespace orc {

struct Stamp {
  int64_t sec;
  int64_t nsec;
};
inline double stamp_diff_sec(const Stamp& a, const Stamp& b) {  // (a - b).toSec()
  int64_t s = a.sec - b.sec, n = a.nsec - b.nsec;
  while (n < 0) { n += 1000000000LL; s -= 1; }
  while (n >= 1000000000LL) { n -= 1000000000LL; s += 1; }
  return (double)s + 1e-9 * (double)n;
}
inline bool stamp_gt(const Stamp& a, const Stamp& b) { return a.sec > b.sec || (a.sec == b.sec && a.nsec > b.nsec); }

// sloam.cpp:428-440: strict '<' keeps the FIRST index on ties.
inline void closest_stamp(const std::vector<Stamp>& packet, const Stamp& stamp, int& idx, double& diff) {
  idx = -1;
  diff = std::numeric_limits<double>::max();
  for (int i = 0; i < (int)packet.size(); ++i) {
    const double d = std::fabs(stamp_diff_sec(packet[i], stamp));
    if (d < diff) { idx = i; diff = d; }
  }
}

struct RelMeas {
  Stamp stamp;
  int robotIndex;
  bool onlyUseOdom;
  int tag;   // caller's handle (position in the caller's original list)
};
struct RelMeasMatch {
  int tag, index, hostIdx, otherIdx;
};

// sloam.cpp:321-412.  pending is mutated exactly as feasible_relative_meas_for_factors is.
inline void find_relmeas_matches(std::vector<RelMeas>& pending, const std::vector<size_t>& pose_counter,
                                 const std::vector<std::vector<Stamp>>& packets, int host,
                                 std::vector<RelMeasMatch>& matches) {
  const double maxTimeDiff = 0.001;
  const std::vector<Stamp>& hostPk = packets.at(host);
  if (pending.empty()) return;
  for (int i = 0; i < (int)pending.size(); i++) {
    RelMeas m = pending[i];
    if (m.robotIndex == host) throw std::runtime_error("robotIndex should not be the same as hostRobotID");
    if (m.onlyUseOdom) throw std::runtime_error("onlyUseOdom measurements shouldn't get to this function");
    int idxOther, idxHost;
    double dt;
    const size_t pc_other = pose_counter[m.robotIndex];
    closest_stamp(packets.at(m.robotIndex), m.stamp, idxOther, dt);
    if (idxOther == -1 || dt > maxTimeDiff || (size_t)idxOther >= pc_other) continue;
    const size_t pc_host = pose_counter[host];
    closest_stamp(hostPk, m.stamp, idxHost, dt);
    if (idxHost == -1 || dt > maxTimeDiff || (size_t)idxHost >= pc_host) continue;
    matches.push_back({m.tag, i, idxHost, idxOther});
    pending.erase(pending.begin() + i);
    i--;
  }
  for (int i = 0; i < (int)pending.size(); i++) {
    const RelMeas& m = pending[i];
    const size_t pc_obs = pose_counter[m.robotIndex], pc_host = pose_counter[host];
    Stamp s_obs{0, 0}, s_host{0, 0};
    if (pc_obs > 0) s_obs = packets.at(m.robotIndex)[pc_obs - 1];
    if (pc_host > 0) s_host = hostPk[pc_host - 1];
    if (stamp_gt(s_obs, m.stamp) && stamp_gt(s_host, m.stamp)) {
      pending.erase(pending.begin() + i);
      i--;
    }
  }
}

}  // namespace orc
