// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the IMU pre-integration on the tracking path:
//   IMU::Preintegrated::{Initialize, IntegrateNewMeasurement, GetDelta{Rotation,Velocity,Position}}   SF/src/ImuTypes.cc:152-170,186-244,292-316
//   IMU::IntegratedRotation, RightJacobianSO3, NormalizeRotation                                      SF/src/ImuTypes.cc:41-58,95-116
//   IMU::Calib::Set (noise covariances)                                                               SF/src/ImuTypes.cc:403-416
//   Tracking::PreintegrateIMU (sample interpolation at the frame borders)                             SF/src/Tracking.cc:1710-1822
//   Tracking::PredictStateIMU                                                                         SF/src/Tracking.cc:1825-1875
// The reference computes in float with Eigen; here the state is kept in float and every update is evaluated in double and
// rounded, NormalizeRotation (Eigen::JacobiSVD, not in tree) is the polar factor by Newton iteration in double.  Agreement
// with a float implementation is therefore ~1e-6 relative, not bitwise.
// PARITY UNPINNED: the reference has no tests or vectors for these.
#pragma once
#include <vector>

namespace oracle {

struct ImuSample { double t; float a[3], w[3]; };  // IMU::Point
struct ImuBias { float bax = 0, bay = 0, baz = 0, bwx = 0, bwy = 0, bwz = 0; };

struct Preintegrated {
    float dT = 0;
    float dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], avgA[3], avgW[3];
    float C[15 * 15];
    float Nga[6], NgaWalk[6];  // diagonals
    ImuBias b;
    int n_measurements = 0;
    Preintegrated(const ImuBias& b_, float ng, float na, float ngw, float naw);
    void IntegrateNewMeasurement(const float acc[3], const float angVel[3], float dt);
    // the same update evaluated in FLOAT, expression by expression in Eigen's order (imu.cpp "float evaluation"): what the product's
    // tc2li_imu_integrate is held to bit for bit; the double form above stays the accuracy check
    void IntegrateNewMeasurementFloat(const float acc[3], const float angVel[3], float dt);
    void GetDeltaRotation(const ImuBias& b_, float out[9]) const;
    void GetDeltaVelocity(const ImuBias& b_, float out[3]) const;
    void GetDeltaPosition(const ImuBias& b_, float out[3]) const;
};

// Tracking::PreintegrateIMU: integrates the samples between the previous and the current frame into p (and returns how many
// integration steps were made).  samples = mvImuFromLastFrame (already selected by the queue logic, Tracking.cc:1731-1764).
int PreintegrateIMU(const std::vector<ImuSample>& samples, double t_prev, double t_cur, Preintegrated& p, bool float_eval = false);

// Tracking::PredictStateIMU (either branch): state 1 + pre-integration (evaluated at bias b) -> state 2
void PredictStateIMU(const Preintegrated& p, const ImuBias& b, const float Rwb1[9], const float twb1[3], const float Vwb1[3],
                     float Rwb2[9], float twb2[3], float Vwb2[3]);

void NormalizeRotation(const float R[9], float out[9]);
void NormalizeRotationFloat(const float R[9], float out[9]);  // Eigen::JacobiSVD<Matrix3f> restated in float: U V^T

}  // namespace oracle
