// Option set: defaults, config.ini reader (replaces ini_parse + handler, main.c:67-96,129) and
// range checks.  Host-only code.
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfcc_host.h"

namespace mfcc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static char *rstrip(char *s)
{
    char *p = s + strlen(s);
    while (p > s && isspace((unsigned char)p[-1])) *--p = 0;
    return s;
}

static char *lskip(char *s)
{
    while (*s && isspace((unsigned char)*s)) ++s;
    return s;
}

// First occurrence of one of `chars`, or of an inline comment (';' preceded by white space),
// or the terminating NUL -- the scanning rule inih uses with inline comments enabled.
static char *find_chars_or_comment(char *s, const char *chars)
{
    int wasSpace = 0;
    while (*s && (!chars || !strchr(chars, *s)) && !(wasSpace && *s == ';')) {
        wasSpace = isspace((unsigned char)*s);
        ++s;
    }
    return s;
}

// One key.  Returns 1 when accepted, 0 for an unknown section/key pair (main.c:94).
static int apply_key(MFCCConfig *c, const char *section, const char *name, const char *value)
{
    auto is = [&](const char *s, const char *n) { return !strcmp(section, s) && !strcmp(name, n); };
    auto save = [&](const char *v) {
        if (!strcmp(v, "f")) return (int)MFCC_SAVE_F; /* main.c:77 */
        if (!strcmp(v, "n")) return (int)MFCC_SAVE_N; /* main.c:78 */
        if (!strcmp(v, "b")) return (int)MFCC_SAVE_B; /* README.md:133 */
        return (int)MFCC_SAVE_E;                      /* main.c:79: anything else */
    };
    if (is("Frame", "sampleRate")) c->sampleRate = atoi(value);
    else if (is("Frame", "hipassfre")) c->hipassfre = atof(value);
    else if (is("Frame", "lowpassfre")) c->lowpassfre = atof(value);
    else if (is("Frame", "preemphasise")) c->preemphasise = atof(value);
    else if (is("Frame", "zeroMeanSigFlag")) c->zeroMeanSigFlag = atoi(value);
    else if (is("Frame", "wlen")) c->wlen = atoi(value);
    else if (is("Frame", "inc")) c->inc = atoi(value);
    else if (is("Frame", "saveType") || is("IO", "saveType")) c->saveType = save(value);
    else if (is("Frame", "vecNum")) c->vecNum = atoi(value);
    else if (is("Frame", "fileList") || is("IO", "fileList")) {
        strncpy(c->fileList, value, MFCC_MAX_PATH - 1);
        c->fileList[MFCC_MAX_PATH - 1] = 0;
    }
    else if (is("IO", "numThreads")) c->numThreads = atoi(value);
    else if (is("IO", "numDevices")) c->numDevices = atoi(value);
    else if (is("IO", "precision")) c->precision = (!strcmp(value, "f64") || !strcmp(value, "1"));
    else if (is("MFCC", "fbankFlag")) c->fbankFlag = atoi(value);
    else if (is("MFCC", "bankNum")) c->bankNum = atoi(value);
    else if (is("MFCC", "MFCCNum")) c->MFCCNum = atoi(value);
    else if (is("MFCC", "MFCC0thFlag")) c->MFCC0thFlag = atoi(value);
    else if (is("MFCC", "usePower")) c->usePower = atoi(value);
    else if (is("MFCC", "takeLogs")) c->takeLogs = atoi(value);
    else if (is("MFCC", "doubleFFT")) c->doubleFFT = atoi(value);
    else if (is("MFCC", "warpAlpha")) c->warpAlpha = atof(value);
    else if (is("MFCC", "warpLowCut")) c->warpLowCut = atof(value);
    else if (is("MFCC", "warpUpCut")) c->warpUpCut = atof(value);
    else if (is("Regression", "normEnergy")) c->normEnergy = atoi(value);
    else if (is("Others", "energyFlag")) c->energyFlag = atoi(value);
    else if (is("Others", "zeroCrossingFlag")) c->zeroCrossingFlag = atoi(value);
    else if (is("Others", "brightFlag")) c->brightFlag = atoi(value);
    else if (is("Others", "subBandEFlag")) c->subBandEFlag = atoi(value);
    else if (is("Others", "fftLength")) c->fftLength = atoi(value);
    else if (is("Regression", "regreOrder")) c->regreOrder = atoi(value);
    else if (is("Regression", "delwin")) c->delwin = atoi(value);
    else if (is("Regression", "znormFlag")) c->znormFlag = atoi(value);
    else return 0;
    return 1;
}

int validate_config(const MFCCConfig &c)
{
    const int M = eff_mfccnum(c);
    if (c.sampleRate <= 0) return set_error("sampleRate must be > 0"), MFCC_E_CONFIG;
    if (c.wlen < 3) return set_error("wlen must be >= 3 (got %d)", c.wlen), MFCC_E_CONFIG;
    if (c.wlen > (2 << kMaxLog2Nc))
        return set_error("wlen %d needs an FFT longer than %d", c.wlen, 2 << kMaxLog2Nc),
               MFCC_E_CONFIG;
    if (c.inc < 1) return set_error("inc must be >= 1 (got %d)", c.inc), MFCC_E_CONFIG;
    if (c.vecNum < 1 || c.vecNum > 4) return set_error("vecNum must be 1..4"), MFCC_E_CONFIG;
    if (c.bankNum < 1 || c.bankNum > 1024) return set_error("bankNum out of range"), MFCC_E_CONFIG;
    if (M < 1 || M > c.bankNum)
        return set_error("MFCCNum must be 1..bankNum (got %d)", M), MFCC_E_CONFIG;
    if (c.regreOrder < 1 || c.regreOrder > 8)
        return set_error("regreOrder must be 1..8"), MFCC_E_CONFIG;
    if (c.regreOrder > 1 && (c.delwin < 1 || c.delwin > 64))
        return set_error("delwin must be 1..64"), MFCC_E_CONFIG;
    if (c.subBandEFlag < 0 || c.fftLength < 0) return set_error("negative count"), MFCC_E_CONFIG;
    if (c.energyFlag < 0 || c.energyFlag > 1 || c.zeroCrossingFlag < 0 || c.zeroCrossingFlag > 1 ||
        c.brightFlag < 0 || c.brightFlag > 1)
        return set_error("energyFlag / zeroCrossingFlag / brightFlag must be 0 or 1"), MFCC_E_CONFIG;
    int fftN = 2;
    while (c.wlen > fftN) fftN *= 2;
    if (c.doubleFFT) fftN *= 2; // mfcc.c:55-56
    if (fftN > (2 << kMaxLog2Nc)) return set_error("doubleFFT needs an FFT longer than %d", 2 << kMaxLog2Nc), MFCC_E_CONFIG;
    if (!(c.warpAlpha > 0.0)) return set_error("warpAlpha must be > 0 (1.0 = no warping)"), MFCC_E_CONFIG;
    if (c.normEnergy < 0 || c.normEnergy > 1) return set_error("normEnergy must be 0 or 1"), MFCC_E_CONFIG;
    if (c.normEnergy && eff_mfccnum(c) >= MFCCFeatureDim(&c))
        return set_error("normEnergy needs a column behind the coefficients (MFCC0thFlag or energyFlag)"), MFCC_E_CONFIG;
    if (c.subBandEFlag > fftN / 2 || c.fftLength > fftN / 2)
        return set_error("subBandEFlag / fftLength exceed fftN/2 = %d", fftN / 2), MFCC_E_CONFIG;
    if (c.znormFlag && MFCCStep(&c) > kMaxZnormStep) // the z-norm kernel keeps a row's means and deviations in LDS
        return set_error("znormFlag supports rows of at most %d columns (this option set has %d)", kMaxZnormStep, MFCCStep(&c)),
               MFCC_E_CONFIG;
    if (c.precision != MFCC_PREC_F32 && c.precision != MFCC_PREC_F64)
        return set_error("precision must be 0 (f32) or 1 (f64)"), MFCC_E_CONFIG;
    return MFCC_OK;
}

} // namespace mfcc

using namespace mfcc;

extern "C" const char *MFCCLastError(void) { return g_err; }

extern "C" int MFCCConfigDefaults(MFCCConfig *cfg)
{
    if (!cfg) return set_error("cfg is NULL"), MFCC_E_ARG;
    memset(cfg, 0, sizeof *cfg);
    cfg->sampleRate = 16000;
    cfg->hipassfre = -1.0;
    cfg->lowpassfre = -1.0;
    cfg->wlen = 400;
    cfg->inc = 160;
    cfg->vecNum = 1;
    cfg->fbankFlag = 1;
    cfg->bankNum = 40;
    cfg->MFCCNum = 40;
    cfg->regreOrder = 1;
    cfg->delwin = 2;
    cfg->saveType = MFCC_SAVE_E;
    cfg->numThreads = 1;
    cfg->numDevices = 0;
    cfg->usePower = 1; // main.c:168
    cfg->takeLogs = 1;
    cfg->warpAlpha = 1.0;
    cfg->precision = MFCC_PREC_F64; // the reference computes in double; single precision is the opt-in fast mode
    return MFCC_OK;
}

extern "C" int MFCCConfigFromIni(const char *iniPath, MFCCConfig *cfg, int *badLine)
{
    if (badLine) *badLine = 0;
    if (!iniPath || !cfg) return set_error("NULL argument"), MFCC_E_ARG;
    FILE *fp = fopen(iniPath, "r");
    if (!fp) return set_error("Can't load '%s'", iniPath), MFCC_E_INI_OPEN;
    MFCCConfigDefaults(cfg);
    char line[MFCC_MAX_PATH + 200], section[64] = "";
    int lineno = 0, firstBad = 0;
    while (fgets(line, sizeof line, fp)) {
        ++lineno;
        char *s = line;
        if (lineno == 1 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB &&
            (unsigned char)s[2] == 0xBF)
            s += 3; // UTF-8 byte-order mark
        s = lskip(rstrip(s));
        if (*s == ';' || *s == '#' || *s == 0) continue;
        if (*s == '[') {
            char *end = find_chars_or_comment(s + 1, "]");
            if (*end == ']') {
                *end = 0;
                strncpy(section, s + 1, sizeof section - 1);
                section[sizeof section - 1] = 0;
            } else if (!firstBad)
                firstBad = lineno;
            continue;
        }
        char *end = find_chars_or_comment(s, "=:");
        if (*end != '=' && *end != ':') {
            if (!firstBad) firstBad = lineno;
            continue;
        }
        *end = 0;
        char *name = rstrip(s);
        char *value = end + 1;
        char *cmt = find_chars_or_comment(value, NULL);
        if (*cmt) *cmt = 0;
        value = lskip(value);
        rstrip(value);
        if (!apply_key(cfg, section, name, value) && !firstBad) firstBad = lineno;
    }
    fclose(fp);
    if (firstBad) {
        if (badLine) *badLine = firstBad;
        set_error("Unknown input config settings in line %d", firstBad); /* main.c:136 */
        return MFCC_E_INI_KEY;
    }
    if (cfg->fbankFlag) { /* main.c:142-144 */
        cfg->MFCCNum = cfg->bankNum;
        cfg->MFCC0thFlag = 0;
    }
    return MFCC_OK;
}

extern "C" long MFCCNumFrames(long numSamples, int wlen, int inc)
{
    if (inc <= 0 || wlen <= 0 || numSamples < wlen) return 0;
    return (numSamples - (wlen - inc)) / inc; /* main.c:211 */
}

extern "C" int MFCCFeatureDim(const MFCCConfig *c)
{
    if (!c) return 0;
    return eff_mfccnum(*c) + eff_c0(*c) + c->energyFlag + c->zeroCrossingFlag + c->brightFlag +
           c->subBandEFlag + c->fftLength; /* main.c:161,185 */
}

extern "C" int MFCCStep(const MFCCConfig *c)
{
    return c ? MFCCFeatureDim(c) * c->vecNum * c->regreOrder : 0; /* main.c:254 */
}
