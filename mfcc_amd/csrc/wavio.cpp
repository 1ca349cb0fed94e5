// Containers: RIFF/WAVE header walk (replaces loadWAVEFile, call site main.c:190-191), feature
// writers (main.c:272-287 text layouts; cnpy.hpp:88-131,235-264 npy layout; raw binary) and the
// synthetic PCM generator used by the benchmark and tests.  Host-only code.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "mfcc_host.h"

namespace mfcc {

static uint32_t le32(const unsigned char *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint32_t le16(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// Walks the chunk list: "fmt " gives the format, anything else before "data" is skipped
// (README.md:68 allows extra chunks between the 44-byte header and the data chunk).
int wav_parse(const unsigned char *f, size_t len, WavInfo &w)
{
    // RF64 (EBU Tech 3306): same layout, 64-bit sizes in a leading "ds64" chunk, for files whose
    // data would overflow the 32-bit size fields
    const bool rf64 = f && len >= 12 && memcmp(f, "RF64", 4) == 0;
    if (!f || len < 12 || (memcmp(f, "RIFF", 4) != 0 && !rf64) || memcmp(f + 8, "WAVE", 4) != 0)
        return set_error("not a RIFF/WAVE file"), MFCC_E_WAV;
    size_t pos = 12;
    bool haveFmt = false;
    int blockAlign = 0;
    uint64_t dataSize64 = 0;
    while (pos + 8 <= len) {
        const uint32_t sz = le32(f + pos + 4);
        const unsigned char *body = f + pos + 8;
        if (!memcmp(f + pos, "fmt ", 4)) {
            if (sz < 16 || pos + 8 + 16 > len) return set_error("short fmt chunk"), MFCC_E_WAV;
            const uint32_t tag = le16(body);
            if (tag != 1 && tag != 0xFFFE)
                return set_error("WAVE format tag %u is not PCM", tag), MFCC_E_WAV;
            // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real format tag
            if (tag == 0xFFFE && sz >= 40 && pos + 8 + 40 <= len && le16(body + 24) != 1)
                return set_error("extensible WAVE sub-format %u is not PCM", le16(body + 24)), MFCC_E_WAV;
            w.channels = (int)le16(body + 2);
            w.sampleRate = (int)le32(body + 4);
            blockAlign = (int)le16(body + 12);
            const int bits = (int)le16(body + 14);
            w.bytesPerSample = bits / 8;
            if (bits % 8 != 0 || w.bytesPerSample < 1 || w.bytesPerSample > 4)
                return set_error("%d-bit PCM is not supported (8, 16, 24 or 32)", bits), MFCC_E_WAV;
            if (w.channels < 1 || w.channels > 2)
                return set_error("%d channels are not supported (1 or 2)", w.channels), MFCC_E_WAV;
            if (blockAlign != w.channels * w.bytesPerSample)
                return set_error("block align %d != channels x sample size", blockAlign), MFCC_E_WAV;
            haveFmt = true;
        } else if (rf64 && !memcmp(f + pos, "ds64", 4)) {
            if (sz < 24 || pos + 8 + 24 > len) return set_error("short ds64 chunk"), MFCC_E_WAV;
            dataSize64 = (uint64_t)le32(body + 8) | ((uint64_t)le32(body + 12) << 32);
        } else if (!memcmp(f + pos, "data", 4)) {
            if (!haveFmt) return set_error("data chunk before fmt chunk"), MFCC_E_WAV;
            size_t bytes = sz, avail = len - (pos + 8);
            if (rf64 && sz == 0xFFFFFFFFu) bytes = (size_t)dataSize64;
            if (bytes > avail) bytes = avail; // size field that overflowed or was never patched
            w.numSamples = (long)(bytes / (size_t)blockAlign);
            w.dataOffset = pos + 8;
            return MFCC_OK;
        }
        pos += 8 + (size_t)sz + (sz & 1u);
    }
    return set_error("no data chunk"), MFCC_E_WAV;
}

int read_file(const char *path, std::vector<unsigned char> &out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return set_error("open %s failed: %s", path, strerror(errno)), MFCC_E_OPEN;
    fseek(fp, 0, SEEK_END);
    const long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), fp);
    fclose(fp);
    if (got != out.size()) return set_error("short read on %s", path), MFCC_E_OPEN;
    return MFCC_OK;
}

} // namespace mfcc

using namespace mfcc;

extern "C" int MFCCWavInfo(const void *wavFile, size_t wavLen, int *channels, int *sampleRate,
                           int *bytesPerSample, long *numSamples, size_t *dataOffset)
{
    WavInfo w;
    const int rc = wav_parse(static_cast<const unsigned char *>(wavFile), wavLen, w);
    if (rc) return rc;
    if (channels) *channels = w.channels;
    if (sampleRate) *sampleRate = w.sampleRate;
    if (bytesPerSample) *bytesPerSample = w.bytesPerSample;
    if (numSamples) *numSamples = w.numSamples;
    if (dataOffset) *dataOffset = w.dataOffset;
    return MFCC_OK;
}

extern "C" int MFCCWriteFeatures(const char *path, const double *data, long numFrames, int step,
                                 int saveType)
{
    if (!path || (!data && numFrames > 0) || step <= 0 || numFrames < 0)
        return set_error("bad argument to MFCCWriteFeatures"), MFCC_E_ARG;
    const bool text = saveType == MFCC_SAVE_F || saveType == MFCC_SAVE_E;
    FILE *fp = fopen(path, text ? "w" : "wb");
    if (!fp) return set_error("open %s failed: %s", path, strerror(errno)), MFCC_E_WRITE;
    const size_t n = (size_t)numFrames * (size_t)step;
    bool ok = true;
    if (text) {
        // "%f\t" / "%e\t" per value, "\n" after every `step` values (main.c:276-285)
        // formatted by fmt_f6 / fmt_e6 (printf-identical, several times faster than fprintf)
        const bool isF = saveType == MFCC_SAVE_F;
        std::vector<char> buf((1 << 20) + 1024);
        size_t used = 0;
        int col = 0;
        for (size_t i = 0; i < n; ++i) {
            used += (size_t)(isF ? fmt_f6(data[i], buf.data() + used) : fmt_e6(data[i], buf.data() + used));
            buf[used++] = '\t';
            if (++col == step) {
                buf[used++] = '\n';
                col = 0;
            }
            if (used > (1 << 20)) {
                ok = ok && fwrite(buf.data(), 1, used, fp) == used;
                used = 0;
            }
        }
        if (used) ok = ok && fwrite(buf.data(), 1, used, fp) == used;
    } else {
        if (saveType == MFCC_SAVE_N) {
            // npy v1.0 header as create_npy_header<double> builds it (cnpy.hpp:235-264)
            std::string dict = "{'descr': '<f8', 'fortran_order': False, 'shape': (";
            dict += std::to_string(numFrames) + ", " + std::to_string(step) + "), }";
            const int remainder = 16 - (int)((10 + dict.size()) % 16);
            dict.append((size_t)remainder, ' ');
            dict.back() = '\n';
            unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, 0, 0};
            pre[8] = (unsigned char)(dict.size() & 0xff);
            pre[9] = (unsigned char)((dict.size() >> 8) & 0xff);
            ok = ok && fwrite(pre, 1, 10, fp) == 10;
            ok = ok && fwrite(dict.data(), 1, dict.size(), fp) == dict.size();
        }
        if (n) ok = ok && fwrite(data, sizeof(double), n, fp) == n;
    }
    if (fclose(fp) != 0) ok = false;
    if (!ok) return set_error("write to %s failed", path), MFCC_E_WRITE;
    return MFCC_OK;
}

// Integer-only generator: u <- u * 6364136223846793005 + 1442695040888963407 (mod 2^64);
// sample = (((u >> 40) & 0xFFFF) + ((u >> 20) & 0xFFFF)) / 2 - 32768 (triangular pdf), shifted
// left 8 for 24-bit, plus an integer-table tone at -12 dBFS.
extern "C" int MFCCSynthPcm(void *dst, int channels, int ch, int bytesPerSample, int64_t numSamples,
                            uint64_t seed, int sampleRate, int toneHz)
{
    return MFCCSynthPcmAt(dst, channels, ch, bytesPerSample, 0, numSamples, seed, sampleRate, toneHz);
}

// Samples firstSample .. firstSample + numSamples - 1 of the same stream, written from dst on: the
// generator's state after firstSample steps is u0 a^k + c (a^k - 1) / (a - 1), found by doubling,
// so ranges of one long stream can be made independently (by threads, or by the ranks of a job).
extern "C" int MFCCSynthPcmAt(void *dst, int channels, int ch, int bytesPerSample, int64_t firstSample,
                              int64_t numSamples, uint64_t seed, int sampleRate, int toneHz)
{
    if (!dst || channels < 1 || ch < 0 || ch >= channels || (bytesPerSample != 2 && bytesPerSample != 3) || firstSample < 0)
        return set_error("bad argument to MFCCSynthPcm"), MFCC_E_ARG;
    // One period of a sine in 1/1024 turns, amplitude 8192 (-12 dBFS of 16-bit), built with
    // integer arithmetic only: Bhaskara's approximation sin(x) ~ 16x(pi-x)/(5pi^2-4x(pi-x)).
    static int16_t table[1024];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 512; ++i) {
            const int64_t x = i, p = 512; // half turn = 512 steps
            const int64_t num = 16 * x * (p - x) * 8192, den = 5 * p * p - 4 * x * (p - x);
            table[i] = (int16_t)(num / den);
            table[i + 512] = (int16_t)(-(num / den));
        }
        init = true;
    }
    unsigned char *out = static_cast<unsigned char *>(dst);
    const size_t stride = (size_t)channels * bytesPerSample;
    uint64_t u = seed;
    {
        // (mul, add) of k steps at once: composing a step with itself doubles k
        uint64_t accMul = 1, accAdd = 0, curMul = 6364136223846793005ull, curAdd = 1442695040888963407ull;
        for (uint64_t k = (uint64_t)firstSample; k; k >>= 1) {
            if (k & 1) {
                accMul *= curMul;
                accAdd = accAdd * curMul + curAdd;
            }
            curAdd = (curMul + 1) * curAdd;
            curMul *= curMul;
        }
        u = u * accMul + accAdd;
    }
    const uint64_t dphase = toneHz > 0 ? (((uint64_t)toneHz << 32) / (uint64_t)sampleRate) : 0;
    uint64_t phase = (dphase * (uint64_t)firstSample) & 0xFFFFFFFFull; // in 1/2^32 turns
    for (int64_t i = 0; i < numSamples; ++i) {
        u = u * 6364136223846793005ull + 1442695040888963407ull;
        int32_t v = (int32_t)((((u >> 40) & 0xFFFF) + ((u >> 20) & 0xFFFF)) >> 1) - 32768;
        if (toneHz > 0) {
            v = v / 2 + table[(phase >> 22) & 1023];
            phase = (phase + dphase) & 0xFFFFFFFFull;
        }
        if (v > 32767) v = 32767;
        if (v < -32768) v = -32768;
        unsigned char *p = out + (size_t)i * stride + (size_t)ch * bytesPerSample;
        if (bytesPerSample == 2) {
            p[0] = (unsigned char)(v & 0xff);
            p[1] = (unsigned char)((v >> 8) & 0xff);
        } else {
            const int32_t w = v * 256 + (int32_t)((u >> 12) & 0xff); // fill the low byte too
            p[0] = (unsigned char)(w & 0xff);
            p[1] = (unsigned char)((w >> 8) & 0xff);
            p[2] = (unsigned char)((w >> 16) & 0xff);
        }
    }
    return MFCC_OK;
}
