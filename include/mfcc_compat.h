/* mfcc_compat.h -- the reference's lower-level API (mfcc.h:29-74) with its own names, signatures and
 * 1-based Vector convention, computed on the GPU (double precision).  Link libmfcc_compat.so (which
 * pulls in libmfcc_amd.so).  A caller that used mfcc.c's pieces one frame at a time keeps its code;
 * every call copies its vectors to the device, runs one kernel and copies the results back, so it
 * costs tens of microseconds: batch work belongs on MFCCWave2FBankDevice / MFCCExtractDevice.
 *
 * There is no CPU arithmetic behind these functions beyond the one-time table set-up of InitFBank
 * (which the reference also does once per run on the host); without an MI355X they print the
 * library's error text and leave their outputs untouched.
 *
 * Vector layout (the hmath convention the reference relies on, mfcc.c:133,185,217): a Vector is a
 * double* whose element [0] holds the length as an int and whose data are elements [1..n]; an int
 * vector holds its length in [0] the same way.  The reference's hmath.h is not part of its tree:
 * define MFCC_COMPAT_HAVE_HMATH before including this header to take Vector / ShortVec from your own
 * hmath.h instead of the typedefs below.
 */
#ifndef MFCC_COMPAT_H
#define MFCC_COMPAT_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef MFCC_COMPAT_HAVE_HMATH
#include "hmath.h"
#else
typedef double *Vector;
typedef int *ShortVec; /* filled through CreateIntVec in the reference (mfcc.c:96) */
#endif

/* mfcc.h:14-27, field for field */
typedef struct {
    int frameSize;
    int numChans;
    double sampPeriod;
    int fftN;
    int klo, khi;
    int usePower;
    int takeLogs;
    double fres;
    Vector cf;       /* [1..numChans+1] */
    ShortVec loChan; /* [1..fftN/2] */
    Vector loWt;     /* [1..fftN/2] */
    Vector x;        /* [1..fftN]: Wave2FBank leaves the packed spectrum here (main.c:235-236 reads it) */
} FBankInfo;

double Mel(int k, double fres);                                              /* mfcc.h:29, mfcc.c:11-15 */
FBankInfo InitFBank(int frameSize, double sampPeriod, int numChans, double lopass, double hipass,
                    int usePower, int takeLogs, int doubleFFT, double alpha, double warpLowCut,
                    double warpUpCut);                                       /* mfcc.h:36-39, mfcc.c:39-122 */
void Wave2FBank(Vector s, Vector fbank, double *te, double *te2, FBankInfo info); /* mfcc.h:44, mfcc.c:125-177 */
void FBank2MFCC(Vector fbank, Vector c, int n);                              /* mfcc.h:52, mfcc.c:180-195 */
double FBank2C0(Vector fbank);                                               /* mfcc.h:58, mfcc.c:197-208 */
void FFT(Vector s, int invert);                                              /* mfcc.h:61, mfcc.c:211-259 */
void Realft(Vector s);                                                       /* mfcc.h:69, mfcc.c:262-291 */

/* Not in the reference (its vectors come from hmath.c): allocation helpers for callers without
 * hmath, and the release of what InitFBank allocated (host vectors and device tables). */
Vector MFCCCreateVector(int n);
ShortVec MFCCCreateIntVec(int n);
int MFCCVectorSize(const void *v);
void MFCCFreeVector(void *v);
void MFCCFreeFBank(FBankInfo *info);
/* 0 if the last call of this thread succeeded, else the library's error code (text: MFCCLastError) */
int MFCCCompatStatus(void);

#ifdef __cplusplus
}
#endif
#endif
