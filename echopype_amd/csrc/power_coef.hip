// K0: per-(channel, ping) coefficient rows for the power-sample calibration, and the time-bin
// CSR used by the binned reductions.  O(C*P) work -- a few microseconds next to the sample
// passes, but kept on the device so that a "step" of the pipeline never leaves HBM.
//
// Reference arithmetic replaced (paths under /root/reference/echopype):
//   calibrate/range.py:138            k = sample_interval * sound_speed / 2
//   calibrate/range.py:174-199        TVG range shift (Ex60: 2 samples; Ex80: c*tau/4; GPT: both)
//   calibrate/cal_params.py:261-324   pulse-length table lookup of gain / sa_correction
//   calibrate/calibrate_ek.py:98,154-162 (CSv), :176-181 (CSp)
//   commongrid/api.py:118-128         pandas-resample bin assignment (left/right closed)
#include "epa_internal.h"

namespace {

struct CoefArgs {
  int C, P, K;
  const double *si, *tau, *pt;
  const double *ss, *ab, *gain, *sa, *pl;
  int ss_mode, ab_mode, gain_mode, sa_mode, psi_mode, taueff_mode;
  const double *psi, *fnom, *taueff;
  const uint8_t* gpt;
  int sonar, cal_type;
  double* coef;
};

__device__ __forceinline__ double fetch(const double* p, int mode, int c, int idx) {
  return mode == EPA_PM_SCALAR ? p[0] : (mode == EPA_PM_CHANNEL ? p[c] : p[idx]);
}

// argmin_k |tau - pulse_length[c,k]|, first minimum, NaN table entries skipped, NaN tau -> NaN
__device__ __forceinline__ double table_lookup(double tau, const double* pl, const double* tab,
                                               int K) {
  if (!(tau == tau)) return __builtin_nan("");
  int best = 0;
  double bestd = __builtin_inf();
  for (int k = 0; k < K; ++k) {
    double d = fabs(tau - pl[k]);
    if (d < bestd) {
      bestd = d;
      best = k;
    }
  }
  return tab[best];
}

__global__ __launch_bounds__(epa::kBlock) void power_coef_ek_kernel(CoefArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.C * a.P) return;
  const int c = idx / a.P;
  const double si = a.si[idx], tau = a.tau[idx], pt = a.pt[idx];
  const double cw = fetch(a.ss, a.ss_mode, c, idx);
  const double alpha = fetch(a.ab, a.ab_mode, c, idx);
  const double G = a.gain_mode == EPA_PM_PULSE_TABLE
                       ? table_lookup(tau, a.pl + (size_t)c * a.K, a.gain + (size_t)c * a.K, a.K)
                       : fetch(a.gain, a.gain_mode, c, idx);
  const double sa = a.sa_mode == EPA_PM_PULSE_TABLE
                        ? table_lookup(tau, a.pl + (size_t)c * a.K, a.sa + (size_t)c * a.K, a.K)
                        : fetch(a.sa, a.sa_mode, c, idx);
  const double k = si * cw / 2;
  const double ex60 = 2 * si * cw / 2;
  // d = shift / k written without the sound speed (it cancels): identical for every ping of a
  // file with constant tau / sample_interval, which is what lets the kernels hoist log10(s - d)
  double shift, d;
  if (a.sonar == EPA_SONAR_EK60) {
    shift = ex60;
    d = 2.0;
  } else {
    shift = cw * tau / 4;
    d = tau / (2 * si);
    if (a.gpt && a.gpt[c]) {
      shift += ex60;
      d += 2.0;
    }
  }
  const double lambda = cw / a.fnom[c];
  const double pi = 3.141592653589793;
  double A;
  if (a.cal_type == EPA_CAL_SV) {
    const double CSv = 10 * log10(pt) + 2 * G + fetch(a.psi, a.psi_mode, c, idx) +
                       10 * log10(lambda * lambda * fetch(a.taueff, a.taueff_mode, c, idx) * cw / (32 * pi * pi));
    A = -CSv - 2 * sa;
  } else {
    const double CSp = 10 * log10(pt) + 2 * G + 10 * log10(lambda * lambda / (16 * pi * pi));
    A = -CSp;
  }
  const double nspread = a.cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  epa::CoefRow r{si, cw / 2, 0.0, shift, 2 * alpha, A + nspread * log10(k), 1.0, d};
  reinterpret_cast<epa::CoefRow*>(a.coef)[idx] = r;
}

// ---- EK80 complex samples: the per-(channel, ping) rows consumed by epa_sv_complex / epa_sv_complex_fft
struct CCoefArgs {
  int C, P, B, bb, cal_type;
  const double* p[EPA_CCP_COUNT];
  int mode[EPA_CCP_COUNT];
  const double* taueff;
  int taueff_mode;
  const uint8_t* gpt;
  double* ccoef;
};

__global__ __launch_bounds__(epa::kBlock) void complex_coef_ek80_kernel(CCoefArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.C * a.P) return;
  const int c = idx / a.P;
  auto get = [&](int k) { return fetch(a.p[k], a.mode[k], c, idx); };
  const double si = get(EPA_CCP_SAMPLE_INTERVAL), tau = get(EPA_CCP_TAU_NOMINAL), pt = get(EPA_CCP_TRANSMIT_POWER);
  const double cw = get(EPA_CCP_SOUND_SPEED), alpha = get(EPA_CCP_ABSORPTION), fc = get(EPA_CCP_FREQ_CENTER);
  const double z_er = get(EPA_CCP_Z_ER), z_et = get(EPA_CCP_Z_ET);
  double gain = get(EPA_CCP_GAIN);
  if (a.bb) {  // transceiver gain compensation of the broadband mode (calibrate_ek.py:507-530)
    const double fa0 = fabs(-get(EPA_CCP_ANGLE_OFFSET_ALONGSHIP)) / (get(EPA_CCP_BEAMWIDTH_ALONGSHIP) / 2);
    const double ft0 = fabs(-get(EPA_CCP_ANGLE_OFFSET_ATHWARTSHIP)) / (get(EPA_CCP_BEAMWIDTH_ATHWARTSHIP) / 2);
    const double fa = fa0 * fa0, ft = ft0 * ft0;
    const double Bt = 0.5 * 6.0206 * (fa + ft - 0.18 * fa * ft);
    gain -= (Bt == Bt) ? Bt : 0.0;  // NaN (missing angles) -> no compensation, as the reference's fillna(0)
  }
  const double lambda = cw / fc;
  double shift = cw * tau / 4;                       // range.py:180-199
  if (a.gpt && a.gpt[c]) shift += 2 * si * cw / 2;
  const double pi = 3.141592653589793;
  double A;
  if (a.cal_type == EPA_CAL_SV) {                    // calibrate_ek.py:610-638
    A = -10 * log10(lambda * lambda * pt * cw / (32 * pi * pi)) - 2 * gain - 10 * log10(fetch(a.taueff, a.taueff_mode, c, idx)) - get(EPA_CCP_PSI);
    if (!a.bb) A -= 2 * get(EPA_CCP_SA_CORRECTION);
  } else {
    A = -10 * log10(lambda * lambda * pt / (16 * pi * pi)) - 2 * gain;
  }
  // prx = B |mean|^2 / (2 sqrt 2)^2 (|z_er + z_et| / z_er)^2 / z_et   (calibrate_ek.py:483-490)
  const double zr = fabs(z_er + z_et) / z_er;
  double* row = a.ccoef + (size_t)idx * EPA_NCCOEF;
  row[EPA_CC_RA] = si;
  row[EPA_CC_RB] = cw / 2;
  row[EPA_CC_SHIFT] = shift;
  row[EPA_CC_ALPHA2] = 2 * alpha;
  row[EPA_CC_A] = A;
  row[EPA_CC_PSCALE] = (double)a.B / 8.0 * (zr * zr) / z_et;
  row[EPA_CC_RSV0] = 0.0;
  row[EPA_CC_RSV1] = 0.0;
}

__global__ __launch_bounds__(epa::kBlock) void pulse_table_lookup_kernel(const double* __restrict__ tau,
                                                                         const double* __restrict__ pl,
                                                                         const double* __restrict__ tab, int C, int P,
                                                                         int K, double* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * P) return;
  const int c = idx / P;
  out[idx] = table_lookup(tau[idx], pl + (size_t)c * K, tab + (size_t)c * K, K);
}

__global__ __launch_bounds__(epa::kBlock) void time_bin_offsets_kernel(const int64_t* t, int P,
                                                                       int64_t t0, int64_t dt,
                                                                       int n_bins,
                                                                       bool closed_right,
                                                                       int32_t* bin_start) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_bins) return;
  const int64_t edge = t0 + (int64_t)b * dt;
  int lo = 0, hi = P;  // first index with t >= edge (left closed) or t > edge (right closed)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int64_t v = t[mid];
    const bool before = closed_right ? (v <= edge) : (v < edge);
    if (before) lo = mid + 1;
    else hi = mid;
  }
  bin_start[b] = lo;
}

}  // namespace

extern "C" int epa_power_coef_ek(int C, int P, const double* sample_interval,
                                 const double* tau_nominal, const double* transmit_power,
                                 const double* sound_speed, int ss_mode, const double* absorption,
                                 int abs_mode, const double* gain, int gain_mode,
                                 const double* sa_correction, int sa_mode,
                                 const double* pulse_length, int K, const double* psi, int psi_mode,
                                 const double* f_nominal, const double* tau_eff, int tau_eff_mode,
                                 const uint8_t* gpt, int sonar, int cal_type, double* coef, epa_stream_t stream) {
  EPA_CHECK_ARG(C > 0 && P > 0, "epa_power_coef_ek: C=%d P=%d must be positive", C, P);
  EPA_CHECK_ARG(sample_interval && tau_nominal && transmit_power && sound_speed && absorption &&
                    gain && sa_correction && psi && f_nominal && tau_eff && coef,
                "epa_power_coef_ek: NULL array argument");
  EPA_CHECK_ARG(ss_mode >= 0 && ss_mode <= 2 && abs_mode >= 0 && abs_mode <= 2,
                "epa_power_coef_ek: bad sound_speed/absorption mode");
  EPA_CHECK_ARG(gain_mode >= 0 && gain_mode <= 3 && sa_mode >= 0 && sa_mode <= 3,
                "epa_power_coef_ek: bad gain/sa mode");
  EPA_CHECK_ARG(psi_mode >= 0 && psi_mode <= 2, "epa_power_coef_ek: bad equivalent_beam_angle mode");
  EPA_CHECK_ARG(tau_eff_mode == EPA_PM_CHANNEL || tau_eff_mode == EPA_PM_CHANNEL_PING,
                "epa_power_coef_ek: tau_eff is per channel or per (channel, ping)");
  if (gain_mode == EPA_PM_PULSE_TABLE || sa_mode == EPA_PM_PULSE_TABLE)
    EPA_CHECK_ARG(pulse_length != nullptr && K > 0,
                  "epa_power_coef_ek: pulse-table mode needs pulse_length and K > 0");
  EPA_CHECK_ARG(sonar == EPA_SONAR_EK60 || sonar == EPA_SONAR_EK80, "epa_power_coef_ek: bad sonar");
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_power_coef_ek: bad cal_type");
  CoefArgs a{C, P, K, sample_interval, tau_nominal, transmit_power, sound_speed, absorption, gain,
             sa_correction, pulse_length, ss_mode, abs_mode, gain_mode, sa_mode, psi_mode, tau_eff_mode, psi, f_nominal,
             tau_eff, gpt, sonar, cal_type, coef};
  const long long n = (long long)C * P;
  const int grid = (int)((n + epa::kBlock - 1) / epa::kBlock);
  hipLaunchKernelGGL(power_coef_ek_kernel, dim3(grid), dim3(epa::kBlock), 0, (hipStream_t)stream, a);
  return epa::check_launch("power_coef_ek_kernel");
}

extern "C" int epa_complex_coef_ek80(int C, int P, const double* const* params, const int* modes,
                                     const double* tau_eff, int tau_eff_mode, const uint8_t* gpt, int B, int bb,
                                     int cal_type, double* ccoef, epa_stream_t stream) {
  EPA_CHECK_ARG(C > 0 && P > 0 && B > 0, "epa_complex_coef_ek80: C=%d P=%d B=%d must be positive", C, P, B);
  EPA_CHECK_ARG(params && modes && tau_eff && ccoef, "epa_complex_coef_ek80: NULL array argument");
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_complex_coef_ek80: bad cal_type");
  EPA_CHECK_ARG(tau_eff_mode == EPA_PM_CHANNEL || tau_eff_mode == EPA_PM_CHANNEL_PING,
                "epa_complex_coef_ek80: tau_eff is per channel or per (channel, ping)");
  CCoefArgs a{};
  a.C = C; a.P = P; a.B = B; a.bb = bb ? 1 : 0; a.cal_type = cal_type;
  for (int k = 0; k < EPA_CCP_COUNT; ++k) {
    const bool needed = !(k >= EPA_CCP_ANGLE_OFFSET_ALONGSHIP && !bb) && !(k == EPA_CCP_SA_CORRECTION && (bb || cal_type != EPA_CAL_SV)) &&
                        !(k == EPA_CCP_PSI && cal_type != EPA_CAL_SV);
    EPA_CHECK_ARG(!needed || params[k] != nullptr, "epa_complex_coef_ek80: parameter %d is NULL", k);
    EPA_CHECK_ARG(modes[k] >= EPA_PM_SCALAR && modes[k] <= EPA_PM_CHANNEL_PING, "epa_complex_coef_ek80: bad mode of parameter %d", k);
    a.p[k] = params[k];
    a.mode[k] = modes[k];
  }
  a.taueff = tau_eff; a.taueff_mode = tau_eff_mode; a.gpt = gpt; a.ccoef = ccoef;
  const long long n = (long long)C * P;
  hipLaunchKernelGGL(complex_coef_ek80_kernel, dim3((unsigned)((n + epa::kBlock - 1) / epa::kBlock)), dim3(epa::kBlock), 0,
                     (hipStream_t)stream, a);
  return epa::check_launch("complex_coef_ek80_kernel");
}

extern "C" int epa_pulse_table_lookup(const double* tau_nominal, const double* pulse_length, const double* table, int C,
                                      int P, int K, double* out, epa_stream_t stream) {
  EPA_CHECK_ARG(tau_nominal && pulse_length && table && out, "epa_pulse_table_lookup: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && K > 0, "epa_pulse_table_lookup: C=%d P=%d K=%d must be positive", C, P, K);
  const long long n = (long long)C * P;
  const int grid = (int)((n + epa::kBlock - 1) / epa::kBlock);
  hipLaunchKernelGGL(pulse_table_lookup_kernel, dim3(grid), dim3(epa::kBlock), 0, (hipStream_t)stream, tau_nominal,
                     pulse_length, table, C, P, K, out);
  return epa::check_launch("pulse_table_lookup_kernel");
}

extern "C" int epa_time_bin_offsets(const int64_t* ping_time, int P, int64_t t0, int64_t dt,
                                    int n_bins, unsigned flags, int32_t* bin_start,
                                    epa_stream_t stream) {
  EPA_CHECK_ARG(ping_time && bin_start, "epa_time_bin_offsets: NULL array argument");
  EPA_CHECK_ARG(P >= 0 && n_bins > 0 && dt > 0, "epa_time_bin_offsets: P=%d n_bins=%d dt=%lld", P,
                n_bins, (long long)dt);
  const int grid = (n_bins + 1 + epa::kBlock - 1) / epa::kBlock;
  hipLaunchKernelGGL(time_bin_offsets_kernel, dim3(grid), dim3(epa::kBlock), 0,
                     (hipStream_t)stream, ping_time, P, t0, dt, n_bins,
                     (flags & EPA_BIN_CLOSED_RIGHT) != 0, bin_start);
  return epa::check_launch("time_bin_offsets_kernel");
}
