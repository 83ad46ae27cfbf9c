// sono_measure_dev.h -- the measurement stage on the GPU (sono_measure_dev.hip): a batch of calls whose dB matrices
// lie in HBM, one wave per call.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sonostft.h"

namespace sono {

// One call of the batch.
struct MeasureJob {
    unsigned long long db_row;          // row of the call's first frame in the batch's dB matrix
    unsigned long long start_frame;     // the call's first frame in its region: the time axis starts there
    unsigned int frames;                // frames of the call
    unsigned int times;                 // centre times the reference hands over with them: frames, or frames + 1 (js:787)
    unsigned int roi;                   // the call's region: row of the zonal table
    unsigned int pad;
    double duration_ms;                 // as the framing stage set it
};

struct MeasureArgs {
    const float *db;                    // the batch's dB matrix, rows of `bins` entries
    unsigned int bins;
    const float *freq_bins;             // bins centre frequencies, Hz
    const double *zonal;                // noise floors, zone_stride entries per region of which `zones` are used
    unsigned int zones, zone_stride;
    const MeasureJob *jobs;
    unsigned long long n_jobs;
    double hop, half_n, sample_rate;    // centre time of frame f: (f * hop + half_n) / sample_rate (js:1350-1351)
    double fhigh_khz, freq_res;
    int debug_mode, time_expansion;
    char *scratch;                      // waves * scratch_per_wave bytes
    unsigned long long scratch_per_wave;
    unsigned int max_frames;            // no call of the batch has more frames
    sono_call_measurement *out;         // one per job
};

size_t measure_scratch_per_wave(unsigned int max_frames, unsigned int bins);
// Waves to launch for n_jobs calls on a device with `cus` compute units (a multiple of the waves per block).
unsigned int measure_waves(unsigned long long n_jobs, unsigned int cus);
hipError_t launch_measure_calls(const MeasureArgs &a, unsigned int waves, hipStream_t s);

}  // namespace sono
