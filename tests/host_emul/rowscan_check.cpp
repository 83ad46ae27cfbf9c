// The row scans of the measurement stage (spectrogram_amd/csrc/sono_rowscan.h) behind plain C functions, so that the
// CPU suite can put them beside the scalar loops they stand for.
#include "../../spectrogram_amd/csrc/sono_measure_core.h"
#include "../../spectrogram_amd/csrc/sono_rowscan.h"

extern "C" {

double measure_pow15(double x) { return sono::mcore::pow15(x); }

float rowscan_float_below(double thr) { return sono::float_below(thr); }
long long rowscan_first_gt(const float *p, long long lo, long long hi, float t) { return sono::first_gt(p, lo, hi, t); }
long long rowscan_last_gt(const float *p, long long lo, long long hi, float t) { return sono::last_gt(p, lo, hi, t); }
float rowscan_row_max(const float *p, unsigned long long n) { return sono::row_max(p, (size_t)n); }
long long rowscan_first_max(const float *p, unsigned long long n, float *value) { return sono::first_max(p, (size_t)n, value); }

}
