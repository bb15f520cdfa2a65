// npz.hpp -- minimal .npz/.npy reader (and a stored-only writer) for the two file formats of
// the MPPI path.  Stands in for cnpy (cnpy::npz_load, used at neural_net_model.cu:82 and
// costs.cu:195 of the reference; cnpy itself is not vendored there and absent in this image).
//
// Supports what numpy.savez / cnpy produce: ZIP local-file entries, method 0 (stored) and, when
// built with -DMPPI_NPZ_ZLIB (links -lz), method 8 (deflate, numpy.savez_compressed); .npy format
// versions 1.0/2.0/3.0, little-endian '<f8', '<f4', '<i4', '<i8', '|u1', C order.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef MPPI_NPZ_ZLIB
#include <zlib.h>
#endif

namespace mppi_host {

struct NpyArray {
  std::vector<size_t> shape;
  char kind = 'f';       // 'f', 'i', 'u', 'b'
  size_t word_size = 0;  // bytes per element
  bool fortran_order = false;
  std::vector<unsigned char> bytes;

  size_t num_vals() const
  {
    size_t n = 1;
    for (size_t s : shape) n *= s;
    return n;
  }
  template <typename T>
  const T *data() const
  {
    return reinterpret_cast<const T *>(bytes.data());
  }
  // element i converted to double, whatever the stored dtype
  double at(size_t i) const
  {
    const unsigned char *p = bytes.data() + i * word_size;
    if (kind == 'f' && word_size == 8) { double v; memcpy(&v, p, 8); return v; }
    if (kind == 'f' && word_size == 4) { float v; memcpy(&v, p, 4); return v; }
    if (kind == 'i' && word_size == 8) { int64_t v; memcpy(&v, p, 8); return (double)v; }
    if (kind == 'i' && word_size == 4) { int32_t v; memcpy(&v, p, 4); return (double)v; }
    if ((kind == 'u' || kind == 'b') && word_size == 1) return (double)*p;
    throw std::runtime_error("npz: unsupported dtype");
  }
};

typedef std::map<std::string, NpyArray> npz_t;

namespace detail {

inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char *p)
{
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

inline NpyArray parse_npy(const unsigned char *p, size_t n)
{
  if (n < 10 || memcmp(p, "\x93NUMPY", 6) != 0) throw std::runtime_error("npz: bad .npy magic");
  const int major = p[6];
  size_t hlen, hoff;
  if (major == 1) { hlen = rd16(p + 8); hoff = 10; }
  else { hlen = rd32(p + 8); hoff = 12; }
  if (hoff + hlen > n) throw std::runtime_error("npz: truncated .npy header");
  const std::string hdr(reinterpret_cast<const char *>(p + hoff), hlen);
  NpyArray a;
  size_t pos = hdr.find("'descr'");
  if (pos == std::string::npos) throw std::runtime_error("npz: no descr");
  pos = hdr.find('\'', hdr.find(':', pos));
  const size_t end = hdr.find('\'', pos + 1);
  const std::string descr = hdr.substr(pos + 1, end - pos - 1);  // e.g. "<f8"
  if (descr.size() < 3 || (descr[0] != '<' && descr[0] != '|' && descr[0] != '='))
    throw std::runtime_error("npz: unsupported byte order in descr " + descr);
  a.kind = descr[1];
  a.word_size = (size_t)std::stoul(descr.substr(2));
  pos = hdr.find("'fortran_order'");
  a.fortran_order = hdr.find("True", pos) != std::string::npos &&
                    hdr.find("True", pos) < hdr.find(',', pos);
  pos = hdr.find("'shape'");
  const size_t lp = hdr.find('(', pos), rp = hdr.find(')', lp);
  std::string sh = hdr.substr(lp + 1, rp - lp - 1);
  size_t i = 0;
  while (i < sh.size()) {
    while (i < sh.size() && (sh[i] < '0' || sh[i] > '9')) i++;
    if (i >= sh.size()) break;
    size_t v = 0;
    while (i < sh.size() && sh[i] >= '0' && sh[i] <= '9') v = v * 10 + (size_t)(sh[i++] - '0');
    a.shape.push_back(v);
  }
  const size_t nbytes = a.num_vals() * a.word_size;
  if (hoff + hlen + nbytes > n) throw std::runtime_error("npz: truncated .npy data");
  a.bytes.assign(p + hoff + hlen, p + hoff + hlen + nbytes);
  return a;
}

inline std::vector<unsigned char> read_file(const std::string &path)
{
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("npz: cannot open " + path);
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> buf((size_t)n);
  if (n > 0 && fread(buf.data(), 1, (size_t)n, f) != (size_t)n) {
    fclose(f);
    throw std::runtime_error("npz: short read " + path);
  }
  fclose(f);
  return buf;
}

}  // namespace detail

inline bool file_exists(const std::string &path)
{
  FILE *f = fopen(path.c_str(), "rb");
  if (f) fclose(f);
  return f != nullptr;  // fileExists, param_getter.cpp:151 region
}

inline npz_t npz_load(const std::string &path)
{
  using namespace detail;
  const std::vector<unsigned char> buf = read_file(path);
  // walk the central directory (sizes in local headers may be zero with data descriptors)
  size_t eocd = std::string::npos;
  for (size_t i = buf.size() >= 22 ? buf.size() - 22 : 0;; i--) {
    if (buf.size() >= 22 && rd32(&buf[i]) == 0x06054b50u) { eocd = i; break; }
    if (i == 0) break;
  }
  if (eocd == std::string::npos) throw std::runtime_error("npz: no end-of-central-directory in " + path);
  const size_t n_entries = rd16(&buf[eocd + 10]);
  size_t cd = rd32(&buf[eocd + 16]);
  npz_t out;
  for (size_t e = 0; e < n_entries; e++) {
    if (cd + 46 > buf.size() || rd32(&buf[cd]) != 0x02014b50u) throw std::runtime_error("npz: bad central dir");
    const uint16_t method = rd16(&buf[cd + 10]);
    uint64_t csize = rd32(&buf[cd + 20]), usize = rd32(&buf[cd + 24]);
    const uint16_t nlen = rd16(&buf[cd + 28]), xlen = rd16(&buf[cd + 30]), clen = rd16(&buf[cd + 32]);
    uint64_t lho = rd32(&buf[cd + 42]);
    std::string name(reinterpret_cast<const char *>(&buf[cd + 46]), nlen);
    // zip64 extra field (numpy writes it when force_zip64 is on)
    size_t x = cd + 46 + nlen;
    const size_t xend = x + xlen;
    while (x + 4 <= xend) {
      const uint16_t id = rd16(&buf[x]), sz = rd16(&buf[x + 2]);
      if (id == 0x0001) {
        size_t q = x + 4;
        auto rd64 = [&](size_t o) { return (uint64_t)rd32(&buf[o]) | ((uint64_t)rd32(&buf[o + 4]) << 32); };
        if (usize == 0xffffffffu) { usize = rd64(q); q += 8; }
        if (csize == 0xffffffffu) { csize = rd64(q); q += 8; }
        if (lho == 0xffffffffu) { lho = rd64(q); q += 8; }
      }
      x += 4 + sz;
    }
    cd = xend + clen;
    if (lho + 30 > buf.size() || rd32(&buf[lho]) != 0x04034b50u) throw std::runtime_error("npz: bad local header");
    const size_t data = lho + 30 + rd16(&buf[lho + 26]) + rd16(&buf[lho + 28]);
    if (data + csize > buf.size()) throw std::runtime_error("npz: truncated entry " + name);
    if (name.size() > 4 && name.substr(name.size() - 4) == ".npy") name.resize(name.size() - 4);
    if (method == 0) {
      out[name] = parse_npy(&buf[data], (size_t)csize);
    } else if (method == 8) {
#ifdef MPPI_NPZ_ZLIB
      std::vector<unsigned char> raw((size_t)usize);
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) throw std::runtime_error("npz: inflateInit2");
      zs.next_in = const_cast<unsigned char *>(&buf[data]);
      zs.avail_in = (uInt)csize;
      zs.next_out = raw.data();
      zs.avail_out = (uInt)usize;
      const int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END) throw std::runtime_error("npz: inflate failed for " + name);
      out[name] = parse_npy(raw.data(), raw.size());
#else
      throw std::runtime_error("npz: entry " + name + " is deflated; rebuild with -DMPPI_NPZ_ZLIB -lz");
#endif
    } else {
      throw std::runtime_error("npz: unsupported zip method for " + name);
    }
  }
  return out;
}

// ---- stored-only writer (tests, synthetic costmaps): float32 / float64 1-D or N-D arrays ----
inline uint32_t crc32_of(const unsigned char *p, size_t n)
{
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

struct NpzWriter {
  struct Entry { std::string name; std::vector<unsigned char> npy; };
  std::vector<Entry> entries;

  void add(const std::string &name, const void *data, const std::vector<size_t> &shape, const char *descr,
           size_t word)
  {
    std::string h = "{'descr': '" + std::string(descr) + "', 'fortran_order': False, 'shape': (";
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); i++) {
      h += std::to_string(shape[i]);
      if (shape.size() == 1 || i + 1 < shape.size()) h += ",";
      if (i + 1 < shape.size()) h += " ";
      n *= shape[i];
    }
    h += "), }";
    while ((10 + h.size() + 1) % 64 != 0) h += ' ';
    h += '\n';
    Entry e;
    e.name = name + ".npy";
    e.npy.assign({0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(h.size() & 0xff),
                  (unsigned char)(h.size() >> 8)});
    e.npy.insert(e.npy.end(), h.begin(), h.end());
    const unsigned char *p = static_cast<const unsigned char *>(data);
    e.npy.insert(e.npy.end(), p, p + n * word);
    entries.push_back(std::move(e));
  }
  void add_f32(const std::string &name, const float *d, const std::vector<size_t> &shape) { add(name, d, shape, "<f4", 4); }
  void add_f64(const std::string &name, const double *d, const std::vector<size_t> &shape) { add(name, d, shape, "<f8", 8); }

  void save(const std::string &path) const
  {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("npz: cannot write " + path);
    auto w16 = [&](uint16_t v) { fputc(v & 0xff, f); fputc(v >> 8, f); };
    auto w32 = [&](uint32_t v) { w16((uint16_t)(v & 0xffff)); w16((uint16_t)(v >> 16)); };
    std::vector<uint32_t> offs, crcs;
    for (const Entry &e : entries) {
      offs.push_back((uint32_t)ftell(f));
      const uint32_t crc = crc32_of(e.npy.data(), e.npy.size());
      crcs.push_back(crc);
      w32(0x04034b50u); w16(20); w16(0); w16(0); w16(0); w16(0x21);
      w32(crc); w32((uint32_t)e.npy.size()); w32((uint32_t)e.npy.size());
      w16((uint16_t)e.name.size()); w16(0);
      fwrite(e.name.data(), 1, e.name.size(), f);
      fwrite(e.npy.data(), 1, e.npy.size(), f);
    }
    const uint32_t cd = (uint32_t)ftell(f);
    for (size_t i = 0; i < entries.size(); i++) {
      const Entry &e = entries[i];
      w32(0x02014b50u); w16(20); w16(20); w16(0); w16(0); w16(0); w16(0x21);
      w32(crcs[i]); w32((uint32_t)e.npy.size()); w32((uint32_t)e.npy.size());
      w16((uint16_t)e.name.size()); w16(0); w16(0); w16(0); w16(0); w32(0); w32(offs[i]);
      fwrite(e.name.data(), 1, e.name.size(), f);
    }
    const uint32_t cdsize = (uint32_t)ftell(f) - cd;
    w32(0x06054b50u); w16(0); w16(0); w16((uint16_t)entries.size()); w16((uint16_t)entries.size());
    w32(cdsize); w32(cd); w16(0);
    fclose(f);
  }
};

}  // namespace mppi_host
