// wav_meta.cpp -- SDR metadata of a WAV capture -> the frequency shift the chain's pre NCO gets.
//
// The step in front of the path for real NRSC-5 captures: the reference's WAV input module reads
// an `auxi` chunk (SDR Console XML, or SDRuno / SDRconnect binary) and the SDR#-style file name, and turns
//   center frequency - --wav-center-target-freq
// into resources->nco_shift_hz, which freq_shift_create prefers over --freq-shift
// (src/input_wav.c:146-432 parsers, 592-629 wav_initialize; src/frequency_shift.c:27-31).
// Host parsing only; here without libsndfile / expat: a RIFF / RF64 chunk walk and a start-tag attribute
// scanner that accepts what the reference's expat handler looks at (<Definition name="value" ...>).
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/iqgpu.h"

namespace {

constexpr size_t kMaxChunk = 1024 * 1024;          // MAX_METADATA_CHUNK_SIZE (include/constants.h)

uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

void copy_str(char *dst, size_t cap, const std::string &v) { snprintf(dst, cap, "%s", v.c_str()); }

// UTC calendar time -> seconds since the epoch (what timegm_portable obtains from mktime under TZ="",
// src/input_wav.c:262-280); mktime's normalisation of out-of-range months / days falls out of the arithmetic
int64_t utc_seconds(int year, int month, int day, int hour, int min, int sec)
{
    int64_t y = year, m = month - 1;
    y += m / 12; m %= 12; if (m < 0) { m += 12; y -= 1; }
    m += 1;
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const int64_t yoe = y - era * 400;
    const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5;          // day of (March-based) year for day 1
    const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    const int64_t days = era * 146097 + doe - 719468 + (day - 1);
    return ((days * 24 + hour) * 60 + min) * 60 + sec;
}

bool decode_entities(std::string &s)
{
    std::string o;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] != '&') { o += s[i]; continue; }
        const size_t e = s.find(';', i);
        if (e == std::string::npos) return false;
        const std::string n = s.substr(i + 1, e - i - 1);
        if (n == "amp") o += '&'; else if (n == "lt") o += '<'; else if (n == "gt") o += '>';
        else if (n == "quot") o += '"'; else if (n == "apos") o += '\'';
        else if (!n.empty() && n[0] == '#') {
            const long c = (n.size() > 1 && (n[1] == 'x' || n[1] == 'X')) ? strtol(n.c_str() + 2, nullptr, 16) : strtol(n.c_str() + 1, nullptr, 10);
            if (c <= 0 || c > 0x7f) o += '?'; else o += (char)c;
        } else return false;
        i = e;
    }
    s = o;
    return true;
}

// expat_start_element_handler, src/input_wav.c:345-412
void definition_attribute(iqgpu_wav_info *md, const std::string &name, const std::string &value)
{
    errno = 0;
    if (name == "SoftwareName") { copy_str(md->software_name, sizeof(md->software_name), value); md->software_name_present = 1; }
    else if (name == "SoftwareVersion") { copy_str(md->software_version, sizeof(md->software_version), value); md->software_version_present = 1; }
    else if (name == "RadioModel") { copy_str(md->radio_model, sizeof(md->radio_model), value); md->radio_model_present = 1; }
    else if (name == "RadioCenterFreq") {
        char *end = nullptr;
        const double d = strtod(value.c_str(), &end);
        if (errno == 0 && end && *end == '\0' && end != value.c_str() && std::isfinite(d)) { md->center_freq_hz = d; md->center_freq_hz_present = 1; }
    } else if (name == "UTCSeconds") {
        if (!md->timestamp_unix_present) {
            char *end = nullptr;
            const long long ts = strtoll(value.c_str(), &end, 10);
            if (errno == 0 && end && *end == '\0' && end != value.c_str()) { md->timestamp_unix = (int64_t)ts; md->timestamp_unix_present = 1; }
        }
    } else if (name == "CurrentTimeUTC") {
        copy_str(md->timestamp_str, sizeof(md->timestamp_str), value); md->timestamp_str_present = 1;
        int year, month, day, hour, min, sec;
        if (sscanf(value.c_str(), "%d-%d-%d %d:%d:%d", &day, &month, &year, &hour, &min, &sec) == 6) {
            md->timestamp_unix = utc_seconds(year, month, day, hour, min, sec); md->timestamp_unix_present = 1;
        }
    }
}

// _parse_auxi_xml_expat, src/input_wav.c:414-438: every <Definition ...> start tag up to the first thing that is
// not well-formed (expat stops there, what it has delivered so far counts)
bool parse_auxi_xml(const unsigned char *data, size_t size, iqgpu_wav_info *md)
{
    const std::string s((const char *)data, size);
    size_t i = 0;
    while (i < s.size()) {
        const size_t lt = s.find('<', i);
        if (lt == std::string::npos) break;
        size_t p = lt + 1;
        if (p >= s.size()) break;
        if (s.compare(p, 3, "!--") == 0) { const size_t e = s.find("-->", p); if (e == std::string::npos) break; i = e + 3; continue; }
        if (s[p] == '?' || s[p] == '!' || s[p] == '/') { const size_t e = s.find('>', p); if (e == std::string::npos) break; i = e + 1; continue; }
        size_t q = p;
        while (q < s.size() && !isspace((unsigned char)s[q]) && s[q] != '>' && s[q] != '/') ++q;
        const std::string tag = s.substr(p, q - p);
        if (tag.empty()) break;
        // attributes
        std::vector<std::pair<std::string, std::string>> atts;
        bool ok = true, closed = false;
        while (q < s.size()) {
            while (q < s.size() && isspace((unsigned char)s[q])) ++q;
            if (q >= s.size()) { ok = false; break; }
            if (s[q] == '>') { ++q; closed = true; break; }
            if (s[q] == '/') { if (q + 1 < s.size() && s[q + 1] == '>') { q += 2; closed = true; } else ok = false; break; }
            size_t n0 = q;
            while (q < s.size() && s[q] != '=' && !isspace((unsigned char)s[q]) && s[q] != '>' && s[q] != '/') ++q;
            const std::string an = s.substr(n0, q - n0);
            while (q < s.size() && isspace((unsigned char)s[q])) ++q;
            if (q >= s.size() || s[q] != '=' || an.empty()) { ok = false; break; }
            ++q;
            while (q < s.size() && isspace((unsigned char)s[q])) ++q;
            if (q >= s.size() || (s[q] != '"' && s[q] != '\'')) { ok = false; break; }
            const char quote = s[q++];
            const size_t v0 = q;
            while (q < s.size() && s[q] != quote) ++q;
            if (q >= s.size()) { ok = false; break; }
            std::string av = s.substr(v0, q - v0);
            ++q;
            if (av.find('<') != std::string::npos || !decode_entities(av)) { ok = false; break; }
            for (auto &prev : atts) if (prev.first == an) ok = false;        // a repeated attribute is not well-formed
            if (!ok) break;
            atts.emplace_back(an, av);
        }
        if (!ok || !closed) break;
        if (tag == "Definition") for (auto &a : atts) definition_attribute(md, a.first, a.second);
        i = q;
    }
    const bool any = md->software_name_present || md->radio_model_present || md->center_freq_hz_present || md->timestamp_unix_present;
    if (any && md->software_name_present && strstr(md->software_name, "SDR Console") != nullptr) md->source_software = IQGPU_SDR_CONSOLE;
    return any;
}

// _parse_binary_auxi_data, src/input_wav.c:282-333: SYSTEMTIME start time, centre frequency as uint32 at byte 32
bool parse_auxi_binary(const unsigned char *data, size_t size, iqgpu_wav_info *md)
{
    if (size < 16 + 16 + 4) return false;
    bool time_parsed = false, freq_parsed = false;
    const unsigned y = rd16(data), mo = rd16(data + 2), d = rd16(data + 6), h = rd16(data + 8), mi = rd16(data + 10), s = rd16(data + 12);
    if (!md->timestamp_unix_present) {
        md->timestamp_unix = utc_seconds((int)y, (int)mo, (int)d, (int)h, (int)mi, (int)s); md->timestamp_unix_present = 1;
        time_parsed = true;
        if (!md->timestamp_str_present) {
            snprintf(md->timestamp_str, sizeof(md->timestamp_str), "%04u-%02u-%02u %02u:%02u:%02u UTC", y, mo, d, h, mi, s);
            md->timestamp_str_present = 1;
        }
    }
    const uint32_t f = rd32(data + 32);
    if (f > 0 && !md->center_freq_hz_present) { md->center_freq_hz = (double)f; md->center_freq_hz_present = 1; freq_parsed = true; }
    return time_parsed || freq_parsed;
}

const char *find_nocase(const char *hay, const char *needle)
{
    for (; *hay; ++hay) {
        const char *h = hay, *n = needle;
        while (*h && *n && tolower((unsigned char)*h) == tolower((unsigned char)*n)) { ++h; ++n; }
        if (!*n) return hay;
    }
    return nullptr;
}

} // namespace

extern "C" void iqgpu_wav_info_init(iqgpu_wav_info *md)
{
    if (!md) return;
    memset(md, 0, sizeof(*md));
    md->source_software = IQGPU_SDR_UNKNOWN;
}

// process_specific_chunk's parse order: XML first, binary if that found nothing (src/input_wav.c:175-181)
extern "C" int iqgpu_wav_parse_auxi(const void *chunk, size_t size, iqgpu_wav_info *md)
{
    if (!chunk || !md || size == 0 || size > kMaxChunk) return 0;
    if (parse_auxi_xml((const unsigned char *)chunk, size, md)) return 1;
    return parse_auxi_binary((const unsigned char *)chunk, size, md) ? 1 : 0;
}

// parse_sdr_metadata_from_filename, src/input_wav.c:192-260
extern "C" int iqgpu_wav_parse_filename(const char *base, iqgpu_wav_info *md)
{
    if (!base || !md) return 0;
    bool parsed = false, sdrsharp = false;
    if (!md->center_freq_hz_present) {
        const char *hz = find_nocase(base, "Hz");
        if (hz) {
            const char *us = nullptr;
            for (const char *t = base; (t = strchr(t, '_')) != nullptr && t < hz; ++t) us = t;
            if (us && us + 1 < hz) {
                const size_t len = (size_t)(hz - (us + 1));
                char num[32];
                if (len < sizeof(num) && len > 0) {
                    memcpy(num, us + 1, len); num[len] = '\0';
                    char *end = nullptr;
                    const double f = strtod(num, &end);
                    if (*end == '\0' && std::isfinite(f) && f > 0) { md->center_freq_hz = f; md->center_freq_hz_present = 1; parsed = true; sdrsharp = true; }
                }
            }
        }
    }
    if (!md->timestamp_unix_present) {
        for (const char *m = strchr(base, '_'); m; m = strchr(m + 1, '_')) {
            int year, month, day, hour, min, sec;
            if (strlen(m) >= 17 && m[9] == '_' && m[16] == 'Z' && sscanf(m, "_%4d%2d%2d_%2d%2d%2dZ", &year, &month, &day, &hour, &min, &sec) == 6) {
                md->timestamp_unix = utc_seconds(year, month, day, hour, min, sec); md->timestamp_unix_present = 1;
                if (!md->timestamp_str_present) {
                    snprintf(md->timestamp_str, sizeof(md->timestamp_str), "%04d-%02d-%02d %02d:%02d:%02d UTC", year, month, day, hour, min, sec);
                    md->timestamp_str_present = 1;
                }
                parsed = true; sdrsharp = true;
                break;
            }
        }
    }
    if (md->source_software == IQGPU_SDR_UNKNOWN) {
        if (sdrsharp) md->source_software = IQGPU_SDR_SHARP;
        else if (strncmp(base, "SDRuno_", 7) == 0) md->source_software = IQGPU_SDR_UNO;
        else if (strncmp(base, "SDRconnect_", 11) == 0) md->source_software = IQGPU_SDR_CONNECT;
        if (md->source_software != IQGPU_SDR_UNKNOWN && !md->software_name_present) {
            static const char *names[] = {"Unknown", "SDR Console", "SDR#", "SDRuno", "SDRconnect"};
            snprintf(md->software_name, sizeof(md->software_name), "%s", names[md->source_software]);
            md->software_name_present = 1;
            parsed = true;
        }
    }
    return parsed ? 1 : 0;
}

// wav_initialize up to the shift (src/input_wav.c:544-629) with the file walked here instead of by libsndfile:
// RIFF / RF64 header, `fmt `, `auxi`, `data`; then the file name.  Returns IQGPU_OK, IQGPU_EINVAL (not a WAV /
// unreadable) or IQGPU_EFORMAT (not 2 channels, or a PCM subtype other than 16-bit signed / 8-bit unsigned).
extern "C" int iqgpu_wav_probe(const char *path, iqgpu_wav_info *md)
{
    if (!path || !md) return IQGPU_EINVAL;
    iqgpu_wav_info_init(md);
    FILE *f = fopen(path, "rb");
    if (!f) return IQGPU_EINVAL;
    unsigned char hdr[12];
    int rc = IQGPU_EINVAL;
    uint64_t file_size = 0;
    if (fseeko(f, 0, SEEK_END) == 0) { const off_t e = ftello(f); if (e > 0) file_size = (uint64_t)e; }
    rewind(f);
    uint64_t ds64_data = 0; bool rf64 = false, have_fmt = false, have_data = false;
    std::vector<unsigned char> auxi;
    if (fread(hdr, 1, 12, f) == 12 && (memcmp(hdr, "RIFF", 4) == 0 || memcmp(hdr, "RF64", 4) == 0) && memcmp(hdr + 8, "WAVE", 4) == 0) {
        rf64 = memcmp(hdr, "RF64", 4) == 0;
        const uint32_t riff_size = rd32(hdr + 4);
        uint64_t pos = 12;
        for (;;) {
            unsigned char ch[8];
            if (fseeko(f, (off_t)pos, SEEK_SET) != 0 || fread(ch, 1, 8, f) != 8) break;
            uint64_t size = rd32(ch + 4);
            const uint64_t body = pos + 8;
            if (memcmp(ch, "ds64", 4) == 0 && size >= 24) {
                unsigned char d[24];
                if (fread(d, 1, 24, f) == 24) ds64_data = rd64(d + 8);
            } else if (memcmp(ch, "fmt ", 4) == 0 && size >= 16) {
                unsigned char d[40]; const size_t n = size < 40 ? (size_t)size : 40;
                if (fread(d, 1, n, f) == n) {
                    int tag = rd16(d);
                    md->channels = rd16(d + 2); md->sample_rate = (int32_t)rd32(d + 4); md->bits_per_sample = rd16(d + 14);
                    if (tag == 0xFFFE && n >= 26) tag = rd16(d + 24);         // WAVE_FORMAT_EXTENSIBLE: sub-format
                    md->format_tag = tag;
                    have_fmt = true;
                }
            } else if (memcmp(ch, "auxi", 4) == 0 && size > 0 && size <= kMaxChunk && auxi.empty()) {
                auxi.resize((size_t)size);
                if (fread(auxi.data(), 1, (size_t)size, f) != (size_t)size) auxi.clear();
            } else if (memcmp(ch, "data", 4) == 0) {
                if (rf64 && size == 0xFFFFFFFFu) size = ds64_data;
                // libsndfile -- what the reference opens the file with (src/input_wav.c:556) -- repairs a capture whose recorder
                // was killed in two ways: a data size of 0 under the 8-byte RIFF size a writer leaves until it closes the file
                // (and more than a bare header on disk) means "to the end of the file" ("wasn't closed properly"), and ANY size
                // beyond the file (0xFFFFFFFF included) is clamped to filelength - dataoffset.  A data chunk that says 0 under
                // a finalised RIFF header is an empty chunk and stays empty: the chunks behind it are not samples.
                const uint64_t avail = file_size > body ? file_size - body : 0;
                if (!rf64 && size == 0 && riff_size == 8 && file_size > 44) size = avail;
                if (size > avail) size = avail;
                md->data_offset = body; md->data_bytes = size; have_data = true;
            }
            pos = body + size + (size & 1);
            if (pos < body) break;
        }
        if (have_fmt && have_data) rc = IQGPU_OK;
    }
    fclose(f);
    if (rc != IQGPU_OK) return rc;
    if (md->channels != 2) return IQGPU_EFORMAT;                              // "must have 2 channels (I/Q)"
    if (md->format_tag == 1 && md->bits_per_sample == 16) md->in_format = IQGPU_FMT_CS16;
    else if (md->format_tag == 1 && md->bits_per_sample == 8) md->in_format = IQGPU_FMT_CU8;
    else return IQGPU_EFORMAT;                                                // unsupported PCM subtype
    if (md->sample_rate <= 0) return IQGPU_EINVAL;
    const size_t bps = iqgpu_get_bytes_per_sample(md->in_format);
    md->frames = md->data_bytes / bps;
    if (!auxi.empty()) md->sdr_info_present = iqgpu_wav_parse_auxi(auxi.data(), auxi.size(), md);
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    if (iqgpu_wav_parse_filename(base, md)) md->sdr_info_present = 1;
    return IQGPU_OK;
}

// the shift rule of wav_initialize, src/input_wav.c:614-629
extern "C" int iqgpu_wav_shift_hz(const iqgpu_wav_info *md, float center_target_hz, float freq_shift_hz_arg, double *nco_shift_hz)
{
    if (!md || !nco_shift_hz) return IQGPU_EINVAL;
    *nco_shift_hz = 0.0;
    if (center_target_hz == 0.0f) return IQGPU_OK;                            // option not used: --freq-shift applies
    if (freq_shift_hz_arg != 0.0f) return IQGPU_ESHIFT;                       // "Conflicting frequency shift options provided"
    if (!md->center_freq_hz_present) return IQGPU_ESHIFT;                     // "does not contain the required center frequency metadata"
    *nco_shift_hz = md->center_freq_hz - (double)center_target_hz;
    return IQGPU_OK;
}
