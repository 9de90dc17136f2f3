// mzml_reader.cpp — mzML input on the host, in C++.
//
// Follows the reference's streaming reader, crates/sage-cloudpath/src/mzml.rs:109-403, for the fields the path uses — the
// same state machine over tag names (spectrum / scan / precursor / selectedIon / binaryDataArray / binary, :153-158, :349-364)
// and the same rules sage_amd/mzml.py implements in Python (kept as the cross-check, tests/test_cli_io.py):
//   * binary arrays: base64, optional zlib (MS:1000574), 32- or 64-bit floats (MS:1000521 / MS:1000523), 64-bit values narrowed
//     to f32 element-wise (:318-326);
//   * numeric cvParam values parsed straight to f32 (`str::parse::<f32>()` and strtof are both correctly rounded);
//   * selected ion m/z (MS:1000744, ignored when 0), charge (MS:1000041); isolation window target as the fallback precursor m/z
//     (MS:1000827, :221-229); isolation_window = Da(-lower, +upper) when both offsets are present (:354-357); a precursor is
//     kept only if its m/z != 0 (:353); the path reads precursors.first();
//   * scan start time in minutes (seconds / 60 in f32, :262-272); inverse reduced ion mobility (MS:1002815);
//   * a spectrum whose total ion current cvParam is 0 is dropped (:205-213); the ms-level filter drops other levels.
// No XML library: mzML's spectrum blocks are flat enough for a tag scanner (attributes in single or double quotes, the five
// predefined entities in attribute values, namespace prefixes stripped).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "host_db.hpp"

namespace sagehip {

namespace {

struct Tag {
    std::string_view name;     // local name
    std::string_view attrs;    // raw attribute text
    bool closing = false, self_closing = false;
    const char* lt = nullptr;  // the tag's '<'
    const char* end = nullptr; // one past '>'
};

// next tag at or after p (comments, processing instructions and CDATA skipped); false at end of input
bool next_tag(const char*& p, const char* e, Tag& t) {
    for (;;) {
        const char* lt = (const char*)std::memchr(p, '<', (size_t)(e - p));
        if (!lt) return false;
        if (e - lt >= 4 && !std::memcmp(lt, "<!--", 4)) {
            const char* c = (const char*)memmem(lt, (size_t)(e - lt), "-->", 3);
            if (!c) return false;
            p = c + 3;
            continue;
        }
        if (e - lt >= 2 && (lt[1] == '?' || lt[1] == '!')) {
            const char* c = (const char*)std::memchr(lt, '>', (size_t)(e - lt));
            if (!c) return false;
            p = c + 1;
            continue;
        }
        const char* q = lt + 1;
        t.lt = lt;
        t.closing = q < e && *q == '/';
        if (t.closing) ++q;
        const char* n0 = q;
        while (q < e && *q != '>' && *q != '/' && !isspace((unsigned char)*q)) ++q;
        std::string_view name(n0, (size_t)(q - n0));
        const size_t colon = name.rfind(':');
        t.name = colon == std::string_view::npos ? name : name.substr(colon + 1);
        const char* a0 = q;
        char quote = 0;
        while (q < e && (quote || *q != '>')) {  // '>' inside a quoted attribute value does not end the tag
            if (quote) {
                if (*q == quote) quote = 0;
            } else if (*q == '"' || *q == '\'') {
                quote = *q;
            }
            ++q;
        }
        if (q >= e) return false;
        t.self_closing = q > a0 && q[-1] == '/';
        t.attrs = std::string_view(a0, (size_t)(q - a0 - (t.self_closing ? 1 : 0)));
        t.end = q + 1;
        p = t.end;
        return true;
    }
}

// value of attribute `key` (exact name match), raw; false when absent
bool attr(std::string_view attrs, std::string_view key, std::string_view& out) {
    size_t i = 0;
    const size_t n = attrs.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)attrs[i])) ++i;
        const size_t k0 = i;
        while (i < n && attrs[i] != '=' && !isspace((unsigned char)attrs[i])) ++i;
        const std::string_view k = attrs.substr(k0, i - k0);
        while (i < n && isspace((unsigned char)attrs[i])) ++i;
        if (i >= n || attrs[i] != '=') return false;
        ++i;
        while (i < n && isspace((unsigned char)attrs[i])) ++i;
        if (i >= n || (attrs[i] != '"' && attrs[i] != '\'')) return false;
        const char quote = attrs[i++];
        const size_t v0 = i;
        while (i < n && attrs[i] != quote) ++i;
        if (i >= n) return false;
        if (k == key) {
            out = attrs.substr(v0, i - v0);
            return true;
        }
        ++i;
    }
    return false;
}

std::string unescape(std::string_view s) {
    std::string o;
    o.reserve(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '&') {
            static const struct { const char* ent; char ch; } kEnt[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool hit = false;
            for (auto& en : kEnt) {
                const size_t l = std::strlen(en.ent);
                if (s.compare(i, l, en.ent) == 0) {
                    o += en.ch;
                    i += l - 1;
                    hit = true;
                    break;
                }
            }
            if (hit) continue;
        }
        o += s[i];
    }
    return o;
}

float parse_f32(std::string_view v) {  // str::parse::<f32>(): one correct rounding; "" -> 0 (mzml.py: _f32)
    if (v.empty()) return 0.0f;
    std::string s(v);
    return std::strtof(s.c_str(), nullptr);
}

void base64_decode(std::string_view in, std::vector<uint8_t>& out) {
    struct Table {
        int8_t v[256];
        Table() {
            std::memset(v, -1, sizeof v);
            const char* al = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
            for (int i = 0; i < 64; ++i) v[(unsigned char)al[i]] = (int8_t)i;
        }
    };
    static const Table table;  // (initialised once, thread-safely: the spectra of a file are decoded in parallel)
    const int8_t* lut = table.v;
    out.clear();
    out.reserve(in.size() * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (unsigned char c : in) {
        if (c == '=') break;
        const int v = lut[c];
        if (v < 0) continue;  // whitespace / foreign characters are skipped, like a lenient decoder
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
}

bool inflate_all(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(in.data());
    zs.avail_in = (uInt)in.size();
    out.resize(std::max<size_t>(in.size() * 4, 1024));
    size_t have = 0;
    int rc;
    do {
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = out.data() + have;
        zs.avail_out = (uInt)(out.size() - have);
        rc = inflate(&zs, Z_NO_FLUSH);
        have = out.size() - zs.avail_out;
    } while (rc == Z_OK);
    inflateEnd(&zs);
    out.resize(have);
    return rc == Z_STREAM_END;
}

// One <spectrum> ... </spectrum> block.  `t` is its opening tag, p the position behind it; on return p is behind the closing
// tag.  false: malformed (err says why; `unterminated`: the input ended inside the block).
struct SpectrumOut {
    std::string id;
    std::vector<float> mz, inten;
    float scan_start = 0.0f, prec_mz = 0.0f, prec_ims = NAN, iso_lo = NAN, iso_hi = NAN;
    uint8_t prec_charge = 0;
    bool have_precursor = false, have_lo = false, have_hi = false, centroid = false, keep = false;
    int level = 0;
};
bool parse_spectrum(Tag t, const char*& p, const char* e, int ms_level, SpectrumOut& o, std::string& err, bool& unterminated) {
    std::vector<uint8_t> raw, plain;
    unterminated = false;
    std::string_view idv;
    std::string id = attr(t.attrs, "id", idv) ? unescape(idv) : std::string();
    bool have_level = false, tic_zero = false, have_precursor = false, in_precursor = false, in_scan = false, in_bda = false;
    bool centroid = false;  // Representation::default() is Profile (spectrum.rs:119-124); MS:1000127 / MS:1000128 set it
    int level = 0, depth = 1;
    std::vector<float> mz, inten;
    float scan_start = 0.0f, prec_mz = 0.0f, prec_ims = NAN, iso_lo = NAN, iso_hi = NAN;
    uint8_t prec_charge = 0;
    float p_mz = 0.0f, p_lo = NAN, p_hi = NAN;  // the precursor being read
    bool p_has_lo = false, p_has_hi = false, have_lo = false, have_hi = false;
    uint8_t p_z = 0;
    int precursor_depth = 0, scan_depth = 0, bda_depth = 0;
    bool bda_f32 = false, bda_zlib = false;
    int bda_kind = 0;  // 1 m/z, 2 intensity
    std::string_view bda_text;
    struct Pending {
        std::string_view text;
        bool f32, zlib, set;
    } pending[2] = {{std::string_view(), false, false, false}, {std::string_view(), false, false, false}};
    bool closed = t.self_closing;
    while (!closed && next_tag(p, e, t)) {
        if (t.closing) {
            if (t.name == "spectrum") {
                closed = true;
                break;
            }
            if (in_precursor && t.name == "precursor" && depth == precursor_depth) {
                in_precursor = false;
                if (p_mz != 0.0f) {  // :353
                    have_precursor = true;
                    prec_mz = p_mz;
                    prec_charge = p_z;
                    iso_lo = p_lo;
                    iso_hi = p_hi;
                    have_lo = p_has_lo;
                    have_hi = p_has_hi;
                }
            } else if (in_scan && t.name == "scan" && depth == scan_depth) {
                in_scan = false;
            } else if (in_bda && t.name == "binaryDataArray" && depth == bda_depth) {
                in_bda = false;
                // (decoded behind the closing </spectrum>, and only when the spectrum passes the level / TIC filters: an MS1
                // survey scan in profile mode is megabytes of base64 + zlib nobody asked for — the reference's parser drops
                // those arrays undecoded, too, mzml.rs:300-330)
                if (!bda_text.empty() && bda_kind) pending[bda_kind - 1] = Pending{bda_text, bda_f32, bda_zlib, true};
            }
            --depth;
            continue;
        }
        // opening (or empty) tag at `depth` + 1
        const int child_depth = depth + 1;
        if (t.name == "cvParam") {
            std::string_view acc, val, unit;
            if (attr(t.attrs, "accession", acc)) {
                const bool has_val = attr(t.attrs, "value", val);
                if (!has_val) val = std::string_view();
                if (in_precursor) {  // every cvParam below <precursor>
                    if (acc == "MS:1000827") {
                        if (p_mz == 0.0f) p_mz = parse_f32(val);
                    } else if (acc == "MS:1000828") {
                        p_lo = parse_f32(val);
                        p_has_lo = true;
                    } else if (acc == "MS:1000829") {
                        p_hi = parse_f32(val);
                        p_has_hi = true;
                    } else if (acc == "MS:1000041") {
                        p_z = (uint8_t)std::atoi(std::string(val).c_str());
                    } else if (acc == "MS:1000744") {
                        const float v = parse_f32(val);
                        if (v != 0.0f) p_mz = v;
                    } else if (acc == "MS:1002815") {
                        prec_ims = parse_f32(val);
                    }
                } else if (in_scan && child_depth == scan_depth + 1) {
                    if (acc == "MS:1000016") {
                        float v = parse_f32(val);
                        if (attr(t.attrs, "unitAccession", unit) && unit == "UO:0000010") v = v / 60.0f;
                        else if (!(unit == "UO:0000031")) {
                            err = "malformed mzML: scan start time unit of spectrum " + id;
                            return false;
                        }
                        scan_start = v;
                    } else if (acc == "MS:1002815") {
                        prec_ims = parse_f32(val);
                    }
                } else if (in_bda && child_depth == bda_depth + 1) {
                    if (acc == "MS:1000514") bda_kind = 1;
                    else if (acc == "MS:1000515" && bda_kind != 1) bda_kind = 2;
                    else if (acc == "MS:1000574") bda_zlib = true;
                    else if (acc == "MS:1000521") bda_f32 = true;
                } else if (child_depth == 2) {  // direct children of <spectrum>
                    if (acc == "MS:1000511") {
                        level = std::atoi(std::string(val).c_str());
                        have_level = true;
                    } else if (acc == "MS:1000285") {
                        tic_zero = parse_f32(val) == 0.0f;
                    } else if (acc == "MS:1000127") {
                        centroid = true;
                    } else if (acc == "MS:1000128") {
                        centroid = false;
                    }
                }
            }
        } else if (t.name == "scan" && !in_scan && !in_precursor) {
            if (!t.self_closing) {
                in_scan = true;
                scan_depth = child_depth;
            }
        } else if (t.name == "precursor" && !in_precursor && !have_precursor) {
            if (!t.self_closing) {
                in_precursor = true;
                precursor_depth = child_depth;
                p_mz = 0.0f;
                p_z = 0;
                p_lo = p_hi = NAN;
                p_has_lo = p_has_hi = false;
            }
        } else if (t.name == "binaryDataArray" && !in_bda) {
            if (!t.self_closing) {
                in_bda = true;
                bda_depth = child_depth;
                bda_f32 = bda_zlib = false;
                bda_kind = 0;
                bda_text = std::string_view();
            }
        } else if (t.name == "binary" && in_bda && child_depth == bda_depth + 1 && !t.self_closing) {
            const char* lt = (const char*)std::memchr(t.end, '<', (size_t)(e - t.end));
            if (!lt) {
                err = "malformed mzML: unterminated <binary>";
                return false;
            }
            bda_text = std::string_view(t.end, (size_t)(lt - t.end));
        }
        if (!t.self_closing) ++depth;
    }
    if (!closed) {
        err = "malformed mzML: unterminated <spectrum>";
        unterminated = true;
        return false;
    }
    o.keep = !(tic_zero || (ms_level >= 0 && (!have_level || level != ms_level)));
    o.id.swap(id);
    if (o.keep) {
        for (int k = 0; k < 2; ++k) {
            if (!pending[k].set) continue;
            base64_decode(pending[k].text, raw);
            const std::vector<uint8_t>* bytes = &raw;
            if (pending[k].zlib) {
                if (!inflate_all(raw, plain)) {
                    err = "malformed mzML: zlib stream of spectrum " + o.id;
                    return false;
                }
                bytes = &plain;
            }
            std::vector<float>& dst = k == 0 ? mz : inten;
            if (pending[k].f32) {
                dst.resize(bytes->size() / 4);
                std::memcpy(dst.data(), bytes->data(), dst.size() * 4);
            } else {
                dst.resize(bytes->size() / 8);
                for (size_t i = 0; i < dst.size(); ++i) {
                    double d;
                    std::memcpy(&d, bytes->data() + 8 * i, 8);
                    dst[i] = (float)d;
                }
            }
        }
        inten.resize(mz.size(), 0.0f);  // (a spectrum with arrays of different lengths is malformed; keep the peak table rectangular)
        o.mz.swap(mz);
        o.inten.swap(inten);
    }
    o.scan_start = scan_start;
    o.prec_mz = prec_mz;
    o.prec_ims = prec_ims;
    o.iso_lo = iso_lo;
    o.iso_hi = iso_hi;
    o.prec_charge = prec_charge;
    o.have_precursor = have_precursor;
    o.have_lo = have_lo;
    o.have_hi = have_hi;
    o.centroid = centroid;
    o.level = level;
    return true;
}

}  // namespace

// MzMLReader::with_file_id_and_level_filter(file_id, ms_level).parse(..): ms_level < 0 keeps every level
bool read_mzml(const char* path, uint32_t file_id, int ms_level, MzmlRun& run, std::string& err) {
    FILE* fh = std::fopen(path, "rb");
    if (!fh) {
        err = std::string("cannot open ") + path;
        return false;
    }
    std::fseek(fh, 0, SEEK_END);
    const long size = std::ftell(fh);
    std::fseek(fh, 0, SEEK_SET);
    std::string text((size_t)std::max<long>(size, 0), '\0');
    const size_t got = size > 0 ? std::fread(&text[0], 1, (size_t)size, fh) : 0;
    std::fclose(fh);
    if ((long)got != size) {
        err = std::string("short read of ") + path;
        return false;
    }
    // gzip-compressed input (sage-cloudpath lib.rs:44-90 gunzips paths ending in gz / gzip; here: by the 1f 8b magic)
    if (text.size() >= 2 && (unsigned char)text[0] == 0x1f && (unsigned char)text[1] == 0x8b) {
        z_stream zs{};
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) {
            err = std::string("zlib initialisation failed for ") + path;
            return false;
        }
        std::string plain_text;
        plain_text.resize(std::max<size_t>(text.size() * 2, 1 << 16));  // (base64 + zlib payload compresses 1.3-3 x; grows geometrically)
        zs.next_in = (Bytef*)text.data();
        zs.avail_in = (uInt)std::min<size_t>(text.size(), 0xFFFFFFFFu);
        size_t consumed_in = zs.avail_in, have = 0;
        int rc = Z_OK;
        for (;;) {
            if (have == plain_text.size()) plain_text.resize(plain_text.size() * 2);
            zs.next_out = (Bytef*)&plain_text[have];
            const size_t room = std::min<size_t>(plain_text.size() - have, 0x40000000u);
            zs.avail_out = (uInt)room;
            rc = inflate(&zs, Z_NO_FLUSH);
            have += room - zs.avail_out;
            if (rc == Z_STREAM_END) {
                if (zs.avail_in == 0 && consumed_in == text.size()) break;
                if (zs.avail_in == 0) break;
                // More input behind a complete member.  Another gzip member (1f 8b: concatenated members, `cat a.gz b.gz`) is
                // inflated like the first, and damage inside it is an error; anything else (zero padding, a signature block)
                // cannot be a member and ends the input.
                if (zs.avail_in >= 2 && zs.next_in[0] == 0x1f && zs.next_in[1] == 0x8b) {
                    if (inflateReset(&zs) != Z_OK) { rc = Z_DATA_ERROR; break; }
                    continue;
                }
                break;  // (rc == Z_STREAM_END: trailing bytes ignored)
            }
            if (rc != Z_OK) break;  // a damaged member — also a second or later one: the spectrum list would be silently truncated
            if (zs.avail_in == 0 && consumed_in < text.size()) {  // (inputs above 4 GiB: feed the next piece)
                const size_t more = std::min<size_t>(text.size() - consumed_in, 0xFFFFFFFFu);
                zs.next_in = (Bytef*)text.data() + consumed_in;
                zs.avail_in = (uInt)more;
                consumed_in += more;
            }
        }
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) {
            err = std::string("corrupt gzip stream in ") + path;
            return false;
        }
        plain_text.resize(have);
        text.swap(plain_text);
    }
    run = MzmlRun{};
    run.peak_off.push_back(0);
    run.id_off.push_back(0);
    // Pass 1, sequential and cheap (a tag scan; base64 text is skipped by memchr): where the top-level <spectrum> blocks are.
    // Pass 2, parallel: every block is decoded on its own (base64, zlib, number parsing — the expensive part) by the same state
    // machine.  Pass 3: the spectra that pass the level / TIC filters are appended in file order.  Errors are reported in file
    // order too: the first malformed block wins, as in a sequential read.
    const char *p = text.data(), *e = text.data() + text.size();
    struct Span {
        const char* begin;  // at or before the '<' of the opening tag
        const char* end;    // behind the closing tag (or the end of the input)
    };
    // top-level <spectrum> blocks from `from` on, until the input ends or — `limit` set — a top-level tag at or behind `limit`
    // shows up; returns that tag's '<' (nullptr: the input ended)
    auto scan_blocks = [&](const char* from, const char* limit, std::vector<Span>& out) -> const char* {
        const char* q = from;
        Tag t;
        for (;;) {
            const char* before = q;
            if (!next_tag(q, e, t)) return nullptr;
            if (limit && t.lt >= limit) return t.lt;
            if (t.closing || t.name != "spectrum") continue;
            bool closed = t.self_closing;
            while (!closed && next_tag(q, e, t)) closed = t.closing && t.name == "spectrum";
            out.push_back(Span{before, closed ? q : e});
            if (!closed) return nullptr;
        }
    };
    std::vector<Span> spans;
    // Large inputs: the scan itself in parallel.  The text is cut at occurrences of the literal `<spectrum ` near equal
    // distances; a piece is scanned from its cut to the next one.  That is the sequential scan exactly if every cut is a
    // top-level position — checked, not assumed: the scanner of piece k must arrive at cut k + 1 as its next top-level tag
    // (it would have run past it inside a block, a comment or a CDATA section otherwise), else the sequential scan runs.
    bool scanned = false;
    size_t piece = (size_t)4 << 20;
    if (const char* v = std::getenv("SAGE_HIP_MZML_PIECE_KB")) piece = (size_t)std::max(1, std::atoi(v)) << 10;  // (tests)
    if (text.size() >= 4 * piece && host_threads() > 1) {
        std::vector<const char*> cuts{p};
        for (size_t at = piece; at < text.size(); at += piece) {
            const char* c = (const char*)memmem(text.data() + at, text.size() - at, "<spectrum ", 10);
            if (!c) break;
            if (c > cuts.back()) cuts.push_back(c);
            at = std::max(at, (size_t)(c - text.data()));
        }
        std::vector<std::vector<Span>> found(cuts.size());
        std::vector<uint8_t> consistent(cuts.size(), 1);
        parallel_for(cuts.size(), 1, [&](size_t b, size_t e2, unsigned) {
            for (size_t k = b; k < e2; k++) {
                const char* limit = k + 1 < cuts.size() ? cuts[k + 1] : nullptr;
                const char* stop = scan_blocks(cuts[k], limit, found[k]);
                // (a piece that runs into the end of the input before its successor's cut swallowed that cut)
                if (limit && stop != limit) consistent[k] = 0;
            }
        });
        scanned = true;
        for (uint8_t c : consistent) scanned = scanned && c;
        if (scanned)
            for (auto& f : found) spans.insert(spans.end(), f.begin(), f.end());
    }
    if (!scanned) {
        spans.clear();
        (void)scan_blocks(p, nullptr, spans);
    }
    std::vector<SpectrumOut> outs(spans.size());
    std::vector<std::string> errs(spans.size());
    std::vector<uint8_t> failed(spans.size(), 0);
    parallel_for(spans.size(), 64, [&](size_t b, size_t e2, unsigned) {
        for (size_t i = b; i < e2; i++) {
            const char* q = spans[i].begin;
            Tag t;
            bool unterminated = false;
            if (!next_tag(q, spans[i].end, t) || !parse_spectrum(t, q, spans[i].end, ms_level, outs[i], errs[i], unterminated)) {
                failed[i] = 1;
                if (errs[i].empty()) errs[i] = "malformed mzML: unterminated <spectrum>";
            }
        }
    });
    for (size_t i = 0; i < spans.size(); i++)
        if (failed[i]) {
            err = errs[i];
            return false;
        }
    // pass 3: offsets of the kept spectra (sequential, cheap), then the peak arrays are copied into place in parallel
    std::vector<size_t> kept;
    kept.reserve(outs.size());
    size_t n_peaks = 0, id_bytes = 0;
    for (size_t i = 0; i < outs.size(); i++)
        if (outs[i].keep) {
            kept.push_back(i);
            run.peak_off.push_back(n_peaks += outs[i].mz.size());
            id_bytes += outs[i].id.size() + 1;
        }
    const size_t n_keep = kept.size();
    run.mz.resize(n_peaks);
    run.intensities.resize(n_peaks);
    run.precursor_mz.resize(n_keep);
    run.precursor_charge.resize(n_keep);
    run.isolation_lo.resize(n_keep);
    run.isolation_hi.resize(n_keep);
    run.scan_start_time.resize(n_keep);
    run.inverse_ion_mobility.resize(n_keep);
    run.file_id.assign(n_keep, file_id);
    run.centroid.resize(n_keep);
    run.has_precursor.resize(n_keep);
    run.ms_level.resize(n_keep);
    parallel_for(n_keep, 256, [&](size_t b, size_t e2, unsigned) {
        for (size_t j = b; j < e2; j++) {
            SpectrumOut& o = outs[kept[j]];
            const size_t at = run.peak_off[j];
            if (!o.mz.empty()) {
                std::memcpy(run.mz.data() + at, o.mz.data(), o.mz.size() * sizeof(float));
                std::memcpy(run.intensities.data() + at, o.inten.data(), o.inten.size() * sizeof(float));
            }
            run.precursor_mz[j] = o.prec_mz;
            run.precursor_charge[j] = o.prec_charge;
            const bool iso = o.have_precursor && o.have_lo && o.have_hi;
            run.isolation_lo[j] = iso ? -o.iso_lo : NAN;
            run.isolation_hi[j] = iso ? o.iso_hi : NAN;
            run.scan_start_time[j] = o.scan_start;
            run.inverse_ion_mobility[j] = o.prec_ims;
            run.centroid[j] = o.centroid ? 1 : 0;
            run.has_precursor[j] = o.have_precursor ? 1 : 0;
            run.ms_level[j] = (uint8_t)std::min(std::max(o.level, 0), 255);
            std::vector<float>().swap(o.mz);
            std::vector<float>().swap(o.inten);
        }
    });
    run.ids.reserve(id_bytes);
    run.id_off.reserve(n_keep + 1);
    for (size_t j = 0; j < n_keep; j++) {
        run.ids += outs[kept[j]].id;
        run.ids += '\0';
        run.id_off.push_back(run.ids.size());
    }
    return true;
}

}  // namespace sagehip
