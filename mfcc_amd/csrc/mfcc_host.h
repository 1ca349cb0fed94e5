// Host-side internals of libmfcc_amd (not part of the C ABI).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mfcc_amd.h"
#include "mfcc_dev.h"

namespace mfcc {

// HTK's hmath.h constants; the reference's hmath.h is not in its tree and README.md:143 says the
// toolbox follows HTK, so tables that the reference builds with `pi` use these values.
constexpr double kHtkPi = 3.14159265358979;
constexpr double kHtkTwoPi = 6.28318530717959;

void set_error(const char *fmt, ...);

// Host tables, all in double.  Index [k-1] holds the reference's 1-based entry [k].
struct Tables {
    int fftN = 0, nc = 0, log2nc = 0, klo = 0, khi = 0;
    double fres = 0, mlo = 0, mhi = 0, mfnorm = 0;
    std::vector<double> cf, loWt, window, dct, tw, splitTw;
    std::vector<int32_t> loChan;
};

int build_tables(const MFCCConfig &cfg, Tables &t);
int build_fbank(int frameSize, double sampPeriod, int numChans, double lopass, double hipass, Tables &t,
                int doubleFFT, double alpha, double warpLowCut, double warpUpCut);
int validate_config(const MFCCConfig &cfg);
inline int eff_mfccnum(const MFCCConfig &c) { return c.fbankFlag ? c.bankNum : c.MFCCNum; }
inline int eff_c0(const MFCCConfig &c) { return c.fbankFlag ? 0 : (c.MFCC0thFlag ? 1 : 0); }

struct WavInfo {
    int channels = 0, sampleRate = 0, bytesPerSample = 0;
    long numSamples = 0;
    size_t dataOffset = 0;
};
int wav_parse(const unsigned char *f, size_t len, WavInfo &w);
int read_file(const char *path, std::vector<unsigned char> &out);

// RCCL, loaded on first use (exchange.hip).  rccl_comm_init_all: one communicator per listed device
// (ncclCommInitAll; the devices must be distinct).
int rccl_load();
int rccl_comm_init_all(void **comms, int n, const int *devices);
void rccl_comm_destroy(void *comm);

// printf-identical "%f" / "%e" of one value into out (>= 400 bytes); returns the length.
int fmt_f6(double x, char *out);
int fmt_e6(double x, char *out);

} // namespace mfcc
