// filters.cpp -- the `--filters` string of `grav1synth diff` (N3): the parser with the reference's grammar and error
// texts (FilterChain::new, /root/reference/src/filters.rs:16-110; its tests :199-363 are the specification), and
// FilterChain::apply (:112-116) for the part this path serves on the device for free:
//
//   crop    removes rows / columns from the SOURCE frame (get_filtered_frame_pair filters the source only,
//           src/main.rs:615-629).  A crop does not touch a sample: it is extent arithmetic on the frame descriptor
//           -- plane pointers move by (top, left), width and height shrink -- for host and device frames alike, so
//           the kernels simply see a smaller frame with the same strides.
//   resize  parsed and validated exactly like the reference, then REFUSED at apply time with a clear error: the
//           reference resamples with the video-resize crate's separable kernels (hermite / catmullrom / mitchell /
//           lanczos / spline36); that arithmetic is a dependency absent from /root/reference and is out of this
//           path's scope: resize the source before `diff`.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/g1s_diff.h"

namespace {

struct Filter {
  uint32_t kind = 0;  // 0 crop, 1 resize
  uint64_t top = 0, bottom = 0, left = 0, right = 0, width = 0, height = 0;
  std::string alg;
};

void set_err(char *err, size_t cap, const std::string &msg) {
  if (!err || !cap) return;
  const size_t n = msg.size() < cap - 1 ? msg.size() : cap - 1;
  std::memcpy(err, msg.data(), n);
  err[n] = 0;
}

// str::parse::<usize>: an optional '+', then decimal digits; the three ParseIntError texts
bool parse_usize(const std::string &v, uint64_t &out, std::string &why) {
  size_t i = 0;
  if (v.empty()) {
    why = "cannot parse integer from empty string";
    return false;
  }
  if (v[0] == '+') i = 1;
  if (i == v.size()) {
    why = "invalid digit found in string";
    return false;
  }
  uint64_t acc = 0;
  for (; i < v.size(); ++i) {
    if (v[i] < '0' || v[i] > '9') {
      why = "invalid digit found in string";
      return false;
    }
    const uint64_t d = (uint64_t)(v[i] - '0');
    if (acc > (UINT64_MAX - d) / 10) {
      why = "number too large to fit in target type";
      return false;
    }
    acc = acc * 10 + d;
  }
  out = acc;
  return true;
}

// str::split: n separators give n + 1 pieces, empty ones included
std::vector<std::string> split(const std::string &s, char sep) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    const size_t b = s.find(sep, a);
    if (b == std::string::npos) {
      out.push_back(s.substr(a));
      return out;
    }
    out.push_back(s.substr(a, b - a));
    a = b + 1;
  }
}

bool split_once(const std::string &s, char sep, std::string &a, std::string &b) {
  const size_t p = s.find(sep);
  if (p == std::string::npos) return false;
  a = s.substr(0, p);
  b = s.substr(p + 1);
  return true;
}

bool parse_chain(const std::string &text, std::vector<Filter> &out, std::string &why) {
  out.clear();
  if (text.empty()) return true;
  for (const std::string &piece : split(text, ';')) {
    std::string name, args;
    if (!split_once(piece, ':', name, args)) {
      why = "Invalid filter syntax in \"" + piece + "\"";
      return false;
    }
    Filter f;
    if (name == "crop") {
      f.kind = 0;
      for (const std::string &arg : split(args, ',')) {
        std::string k, v;
        if (!split_once(arg, '=', k, v)) {
          why = "Invalid filter syntax in \"" + arg + "\"";
          return false;
        }
        uint64_t *dst = k == "top" ? &f.top : k == "bottom" ? &f.bottom : k == "left" ? &f.left : k == "right" ? &f.right : nullptr;
        if (!dst) {
          why = "Unrecognized crop arg \"" + k + "\"";
          return false;
        }
        if (!parse_usize(v, *dst, why)) return false;
      }
    } else if (name == "resize") {
      f.kind = 1;
      f.alg = "catmullrom";
      for (const std::string &arg : split(args, ',')) {
        std::string k, v;
        if (!split_once(arg, '=', k, v)) {
          why = "Invalid filter syntax in \"" + arg + "\"";
          return false;
        }
        if (k == "width" || k == "height") {
          if (!parse_usize(v, k == "width" ? f.width : f.height, why)) return false;
        } else if (k == "alg") {
          if (v != "hermite" && v != "catmullrom" && v != "mitchell" && v != "lanczos" && v != "spline36") {
            why = "Unrecognized resize algorithm \"" + v + "\"";
            return false;
          }
          f.alg = v;
        } else {
          why = "Unrecognized resize arg \"" + k + "\"";
          return false;
        }
      }
      if (f.width == 0 || f.height == 0) {
        why = "Both width and height must be provided to resize filter";
        return false;
      }
    } else {
      why = "Unrecognized filter \"" + name + "\"";
      return false;
    }
    out.push_back(f);
  }
  return true;
}

}  // namespace

struct g1s_filters {
  std::vector<Filter> filters;
};

extern "C" {

g1s_filters_t *g1s_filters_new(const char *text, char *err, size_t errcap) {
  std::vector<Filter> parsed;
  std::string why;
  if (!parse_chain(text ? text : "", parsed, why)) {
    set_err(err, errcap, why);
    return nullptr;
  }
  g1s_filters *f = new g1s_filters;
  f->filters.swap(parsed);
  return f;
}

size_t g1s_filters_len(const g1s_filters_t *f) { return f ? f->filters.size() : 0; }

int g1s_filters_get(const g1s_filters_t *f, size_t i, g1s_filter_desc_t *out) {
  if (!f || !out || i >= f->filters.size()) return G1S_ERR_INVALID;
  const Filter &x = f->filters[i];
  std::memset(out, 0, sizeof(*out));
  out->kind = x.kind;
  out->top = x.top;
  out->bottom = x.bottom;
  out->left = x.left;
  out->right = x.right;
  out->width = x.width;
  out->height = x.height;
  snprintf(out->alg, sizeof(out->alg), "%s", x.alg.c_str());
  return G1S_OK;
}

int g1s_filters_apply(const g1s_filters_t *f, const g1s_frame_t *in, g1s_frame_t *out, char *err, size_t errcap) {
  if (!in || !out) return G1S_ERR_INVALID;
  g1s_frame_t fr = *in;
  if (f) {
    for (const Filter &x : f->filters) {
      if (x.kind == 1) {
        set_err(err, errcap, "resize:width=" + std::to_string(x.width) + ",height=" + std::to_string(x.height) + ",alg=" + x.alg +
                                 " -- the resize filter is not supported here (crop is): resize the source before diff");
        return G1S_ERR_UNSUPPORTED;
      }
      // crop: the frame must keep at least one sample, and with decimated chroma the cut must fall on a chroma sample
      // (each amount on its own: the parser accepts anything up to UINT64_MAX and a sum would wrap)
      if (x.left >= fr.width || x.right >= fr.width - x.left || x.top >= fr.height || x.bottom >= fr.height - x.top) {
        set_err(err, errcap, "crop:top=" + std::to_string(x.top) + ",bottom=" + std::to_string(x.bottom) + ",left=" + std::to_string(x.left) +
                                 ",right=" + std::to_string(x.right) + " leaves nothing of a " + std::to_string(fr.width) + "x" +
                                 std::to_string(fr.height) + " frame");
        return G1S_ERR_INVALID;
      }
      const uint64_t mx = fr.nplanes == 3 ? (1u << fr.xdec) - 1u : 0u, my = fr.nplanes == 3 ? (1u << fr.ydec) - 1u : 0u;
      if ((x.left & mx) || (x.right & mx) || (x.top & my) || (x.bottom & my)) {
        set_err(err, errcap, "crop amounts must be multiples of the chroma subsampling (" + std::to_string(mx + 1) + " horizontally, " +
                                 std::to_string(my + 1) + " vertically)");
        return G1S_ERR_INVALID;
      }
      for (uint32_t c = 0; c < fr.nplanes; ++c) {
        const uint64_t l = c ? x.left >> fr.xdec : x.left, t = c ? x.top >> fr.ydec : x.top;
        fr.data[c] = static_cast<const uint8_t *>(fr.data[c]) + t * fr.stride_bytes[c] + l * fr.bytes_per_sample;
      }
      fr.width -= (uint32_t)(x.left + x.right);
      fr.height -= (uint32_t)(x.top + x.bottom);
    }
  }
  *out = fr;
  return G1S_OK;
}

void g1s_filters_free(g1s_filters_t *f) { delete f; }

}  // extern "C"
