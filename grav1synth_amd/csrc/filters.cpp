// filters.cpp -- the `--filters` string of `grav1synth diff` (N3): the parser with the reference's grammar and error
// texts (FilterChain::new, /root/reference/src/filters.rs:16-110; its tests :199-363 are the specification), and
// FilterChain::apply (:112-116) for the part this path serves on the device for free:
//
//   crop    removes rows / columns from the SOURCE frame (get_filtered_frame_pair filters the source only,
//           src/main.rs:615-629).  A crop does not touch a sample: it is extent arithmetic on the frame descriptor
//           -- plane pointers move by (top, left), width and height shrink -- for host and device frames alike, so
//           the kernels simply see a smaller frame with the same strides.
//   resize  parsed and validated exactly like the reference; applied ON THE DEVICE by resize.hip (the five separable
//           kernels -- hermite / catmullrom / mitchell / lanczos / spline36 -- of the video-resize crate, a dependency absent
//           from /root/reference: parity unpinned, the assumptions are listed there).  The resized frame lives in a ring
//           of device buffers owned by the chain; the frame-pair loop (ingest.cpp) keeps a slot until the generator has
//           released the frame.  A resize needs the source bit depth (FilterChain::apply's `source_bd`):
//           g1s_filters_apply_bd.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <memory>

#include "../../include/g1s_diff.h"
#include "resize.h"

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
  // one state per resize filter of the chain (taps, staging, the ring of output frames); made at the first apply
  mutable std::vector<std::unique_ptr<g1s::ResizeState>> resize;
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

// FilterChain::apply(frame, source_bd) (src/filters.rs:112-116).  device: where a resize runs (-1: the current device);
// slot: which buffer of the resize filters' output rings receives the frame (the caller keeps slot s untouched for as long
// as it uses the frame that went there).  bit_depth 0: only legal for chains without a resize.
int g1s_filters_apply_bd(const g1s_filters_t *f, const g1s_frame_t *in, uint32_t bit_depth, int32_t device, uint32_t slot, g1s_frame_t *out,
                         char *err, size_t errcap) {
  if (!in || !out) return G1S_ERR_INVALID;
  g1s_frame_t fr = *in;
  if (f) {
    if (f->resize.size() < f->filters.size()) f->resize.resize(f->filters.size());
    for (size_t fi = 0; fi < f->filters.size(); ++fi) {
      const Filter &x = f->filters[fi];
      if (x.kind == 1) {
        if (bit_depth == 0) {
          set_err(err, errcap, "resize:width=" + std::to_string(x.width) + ",height=" + std::to_string(x.height) + ",alg=" + x.alg +
                                   " -- a resize needs the source bit depth (g1s_filters_apply_bd)");
          return G1S_ERR_UNSUPPORTED;
        }
        if (x.width > 65535 || x.height > 65535) {
          set_err(err, errcap, "resize: target larger than 65535 x 65535");
          return G1S_ERR_INVALID;
        }
        if (!f->resize[fi]) f->resize[fi].reset(new g1s::ResizeState(g1s::resize_alg_id(x.alg.c_str())));
        std::string why;
        g1s_frame_t resized;
        const int rc = f->resize[fi]->run(fr, bit_depth, (uint32_t)x.width, (uint32_t)x.height, device, (int)slot, resized, why);
        if (rc) {
          set_err(err, errcap, why);
          return rc;
        }
        fr = resized;
        continue;
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

int g1s_filters_apply(const g1s_filters_t *f, const g1s_frame_t *in, g1s_frame_t *out, char *err, size_t errcap) {
  // (8-bit samples say their depth; deeper ones need g1s_filters_apply_bd when the chain resizes)
  return g1s_filters_apply_bd(f, in, in && in->bytes_per_sample == 1 ? 8u : 0u, -1, 0, out, err, errcap);
}

int g1s_filters_has_resize(const g1s_filters_t *f) {
  if (f)
    for (const Filter &x : f->filters)
      if (x.kind == 1) return 1;
  return 0;
}

void g1s_filters_free(g1s_filters_t *f) { delete f; }

}  // extern "C"
