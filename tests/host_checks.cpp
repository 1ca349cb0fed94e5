// Host-only exercise of the library's host code (ini parser, WAV walk, writers, number formatter,
// table builder, PCM generator), meant to be built with -fsanitize=address,undefined (make asan in
// mfcc_amd/csrc) and run by tests/test_host.py: exits 0 when every check passed and the sanitizers
// stayed silent.  No GPU is touched.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/mfcc_amd.h"

static int g_failed = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "host_checks: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                      \
        }                                                                    \
    } while (0)

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return g_rng;
}

static std::vector<unsigned char> wav_image(int channels, int bps, int rate, const std::vector<unsigned char> &pcm, bool extra)
{
    std::vector<unsigned char> f;
    auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) f.push_back((unsigned char)(v >> (8 * i))); };
    auto put16 = [&](uint32_t v) { for (int i = 0; i < 2; ++i) f.push_back((unsigned char)(v >> (8 * i))); };
    f.insert(f.end(), {'R', 'I', 'F', 'F'});
    put32(0);
    f.insert(f.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '});
    put32(16);
    put16(1), put16((uint32_t)channels), put32((uint32_t)rate), put32((uint32_t)(rate * channels * bps));
    put16((uint32_t)(channels * bps)), put16((uint32_t)(8 * bps));
    if (extra) {
        f.insert(f.end(), {'L', 'I', 'S', 'T'});
        put32(5);
        f.insert(f.end(), {'a', 'b', 'c', 'd', 'e', 0}); // odd chunk: one pad byte
    }
    f.insert(f.end(), {'d', 'a', 't', 'a'});
    put32((uint32_t)pcm.size());
    f.insert(f.end(), pcm.begin(), pcm.end());
    const uint32_t riff = (uint32_t)f.size() - 8;
    for (int i = 0; i < 4; ++i) f[4 + i] = (unsigned char)(riff >> (8 * i));
    return f;
}

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    // ---- ini parser: the reference's example layout, README's keys, errors
    {
        const std::string ini = tmp + "/hc.ini";
        FILE *fp = fopen(ini.c_str(), "w");
        fputs("[Frame];\nsampleRate = 44100 ;\nhipassfre = 22050;\nlowpassfre=10\npreemphasise = 0.97\nwlen = 1764\ninc=882\n"
              "saveType = f ; text\nvecNum = 2\nfileList = list.txt\n[MFCC];\nfbankFlag = 1\nbankNum = 40\nMFCCNum = 13\n"
              "MFCC0thFlag = 1\n[Others]\nenergyFlag = 1\nzeroCrossingFlag=1\nbrightFlag = 1\nsubBandEFlag = 8\nfftLength = 0\n"
              "[Regression]\nregreOrder = 3\ndelwin = 9\nznormFlag = 1\n[IO]\nnumThreads = 4\nnumDevices = 2\nprecision = f32\n",
              fp);
        fclose(fp);
        MFCCConfig cfg;
        int bad = -1;
        CHECK(MFCCConfigFromIni(ini.c_str(), &cfg, &bad) == MFCC_OK);
        CHECK(cfg.sampleRate == 44100 && cfg.wlen == 1764 && cfg.vecNum == 2 && cfg.MFCCNum == 40 && cfg.MFCC0thFlag == 0);
        CHECK(cfg.numThreads == 4 && cfg.numDevices == 2 && cfg.precision == MFCC_PREC_F32 && cfg.znormFlag == 1);
        CHECK(MFCCStep(&cfg) == 306 && MFCCFeatureDim(&cfg) == 51);
        fp = fopen(ini.c_str(), "w");
        fputs("[Frame]\nwlen = 5\nbogusKey = 1\n", fp);
        fclose(fp);
        CHECK(MFCCConfigFromIni(ini.c_str(), &cfg, &bad) == MFCC_E_INI_KEY && bad == 3);
        CHECK(MFCCConfigFromIni((tmp + "/no/such.ini").c_str(), &cfg, &bad) == MFCC_E_INI_OPEN);
        // a line longer than any buffer, binary junk
        fp = fopen(ini.c_str(), "w");
        fputs("[Frame]\nfileList = ", fp);
        for (int i = 0; i < 5000; ++i) fputc('x', fp);
        fputs("\n\xff\xfe\x00junk\n", fp);
        fclose(fp);
        (void)MFCCConfigFromIni(ini.c_str(), &cfg, &bad); // any status, no memory error
        CHECK(MFCCNumFrames(1323001, 1764, 882) == 1499 && MFCCNumFrames(1763, 1764, 882) == 0);
    }
    // ---- WAV walk: layouts, extra chunks, truncation, mutated headers
    {
        std::vector<unsigned char> pcm(2 * 3 * 1000);
        for (auto &b : pcm) b = (unsigned char)rnd();
        for (int extra = 0; extra < 2; ++extra) {
            std::vector<unsigned char> f = wav_image(2, 3, 44100, pcm, extra != 0);
            int ch = 0, sr = 0, bps = 0;
            long ns = 0;
            size_t off = 0;
            CHECK(MFCCWavInfo(f.data(), f.size(), &ch, &sr, &bps, &ns, &off) == MFCC_OK);
            CHECK(ch == 2 && sr == 44100 && bps == 3 && ns == 1000 && off + pcm.size() == f.size());
            for (size_t cut = 0; cut < f.size(); cut += 7) { // every truncation: a status, never a fault
                const int rc = MFCCWavInfo(f.data(), cut, &ch, &sr, &bps, &ns, &off);
                if (rc == MFCC_OK) CHECK(off + (size_t)ns * ch * bps <= cut);
            }
            for (int it = 0; it < 20000; ++it) { // random header damage
                std::vector<unsigned char> g(f.begin(), f.begin() + 120);
                for (int k = 0; k < 3; ++k) g[rnd() % 64] = (unsigned char)rnd();
                const int rc = MFCCWavInfo(g.data(), g.size(), &ch, &sr, &bps, &ns, &off);
                if (rc == MFCC_OK) CHECK(ns >= 0 && off + (size_t)ns * ch * bps <= g.size());
            }
        }
    }
    // ---- number formatter against printf, writers
    {
        char a[512], b[512];
        const double special[] = {0.0, -0.0, 1.0, -1.0, 0.5, 9.9999995, 999999.9999995, 1e-7, 5e-324, 1.7976931348623157e308,
                                  123456789.123456789, -2.5e-5, 1e15, 1e16, 1e22, 1e23, INFINITY, -INFINITY, NAN};
        std::vector<double> vals(special, special + sizeof special / sizeof special[0]);
        for (int i = 0; i < 200000; ++i) {
            uint64_t bits = rnd();
            double x;
            memcpy(&x, &bits, 8);
            vals.push_back(x);
            vals.push_back((double)(int64_t)(rnd() % 2000001 - 1000000) / 1024.0);
        }
        for (double x : vals)
            for (int type = 0; type < 2; ++type) {
                const int n = MFCCFormatValue(x, type == 0 ? MFCC_SAVE_F : MFCC_SAVE_E, a, sizeof a);
                snprintf(b, sizeof b, type == 0 ? "%f" : "%e", x);
                CHECK(n == (int)strlen(b) && !strcmp(a, b));
            }
        std::vector<double> m(7 * 5);
        for (size_t i = 0; i < m.size(); ++i) m[i] = (double)i * 1.25 - 3.0;
        const int types[] = {MFCC_SAVE_F, MFCC_SAVE_E, MFCC_SAVE_N, MFCC_SAVE_B};
        for (int t : types) CHECK(MFCCWriteFeatures((tmp + "/hc.out").c_str(), m.data(), 7, 5, t) == MFCC_OK);
        CHECK(MFCCWriteFeatures((tmp + "/hc.out").c_str(), m.data(), 0, 5, MFCC_SAVE_N) == MFCC_OK);
        CHECK(MFCCWriteFeatures((tmp + "/no/such/dir/x").c_str(), m.data(), 7, 5, MFCC_SAVE_F) == MFCC_E_WRITE);
    }
    // ---- host tables (InitFBank + window) for several option sets
    {
        const int cases[][4] = {{1764, 44100, 40, 2048}, {400, 16000, 40, 512}, {400, 16000, 26, 512}, {2, 8000, 1, 2}, {300, 8000, 128, 512}};
        for (const auto &cs : cases) {
            MFCCConfig cfg;
            MFCCConfigDefaults(&cfg);
            cfg.wlen = cs[0], cfg.inc = cs[0] / 2 > 0 ? cs[0] / 2 : 1, cfg.sampleRate = cs[1], cfg.bankNum = cs[2], cfg.fbankFlag = 1;
            cfg.lowpassfre = 10, cfg.hipassfre = cs[1] / 2;
            int fftN = 0, klo = 0, khi = 0;
            char text[256];
            if (MFCCHostTables(&cfg, &fftN, &klo, &khi, nullptr, nullptr, nullptr, nullptr, text, sizeof text) != MFCC_OK) continue;
            CHECK(fftN == cs[3]);
            std::vector<double> cf((size_t)cs[2] + 1), loWt((size_t)fftN / 2), win((size_t)cs[0]);
            std::vector<int> loChan((size_t)fftN / 2);
            CHECK(MFCCHostTables(&cfg, nullptr, nullptr, nullptr, cf.data(), loChan.data(), loWt.data(), win.data(), nullptr, 0) == MFCC_OK);
        }
    }
    // ---- PCM generator: a range equals the slice of the whole stream
    {
        std::vector<unsigned char> whole(3000 * 2 * 2), part(500 * 2 * 2);
        for (int ch = 0; ch < 2; ++ch) {
            CHECK(MFCCSynthPcm(whole.data(), 2, ch, 2, 3000, 42 + ch, 16000, 440) == MFCC_OK);
            CHECK(MFCCSynthPcmAt(part.data(), 2, ch, 2, 1234, 500, 42 + ch, 16000, 440) == MFCC_OK);
        }
        CHECK(!memcmp(part.data(), whole.data() + 1234 * 4, part.size()));
    }
    if (g_failed) fprintf(stderr, "host_checks: %d check(s) failed\n", g_failed);
    else printf("host_checks: ok\n");
    return g_failed ? 1 : 0;
}
