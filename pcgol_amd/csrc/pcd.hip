// pcd.hip -- PCD wire format -> device: header parse, ascii / binary /
// binary_compressed payloads, LZF decode, and the SoA -> AoS de-interleave of compressed files
// as a kernel, so a file's records land in HBM ready for pcgx_voxel_filter_dev & co. without a
// host-side AoS copy.  Also the writer (always "DATA binary").
//
// Reference: pc/io.go:33-45 (Unmarshal), :47-136 (unmarshalPCDHeaderTo), :138-230
// (unmarshalPCDDataTo), :232-285 (Marshal).  LZF: the reference's only third-party module,
// github.com/zhuyie/golzf v0.0.0-20161112031142-8387b0307ade (go.mod:5, a port of liblzf), is not
// part of /root/reference; lzf_decompress below restates the published liblzf stream format.
//
// Quirk kept (io.go:217-226): the de-interleave copies Size[i] bytes of field i from
// head[i] + p*Size[i]; COUNT is ignored on both sides, so for a COUNT > 1 field only element 0 is
// filled (from the first 1/COUNT of the field's block) and the other elements stay zero.
#include <errno.h>
#include <stdlib.h>
#include <limits.h>
#include <string.h>

#include <string>
#include <vector>

#include "pcgx_internal.h"

namespace pcgx {

struct PcdLayout {
  int32_t n_fields;
  int32_t size[PCGX_PCD_MAX_FIELDS];
  int64_t head[PCGX_PCD_MAX_FIELDS];    // start of field i's block in the decoded stream
  int32_t offset[PCGX_PCD_MAX_FIELDS];  // byte offset of field i inside a record
  int64_t stride;
};

// record p, field i: Size[i] bytes from head[i] + p*Size[i]  (io.go:217-226)
__global__ __launch_bounds__(256) void pcd_deinterleave_kernel(const uint8_t *__restrict__ dec, int64_t dec_len,
                                                               int64_t points, PcdLayout lay,
                                                               uint8_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= points * lay.n_fields) return;
  const int64_t p = t / lay.n_fields;
  const int i = (int)(t % lay.n_fields);
  const int size = lay.size[i];
  const int64_t from = lay.head[i] + p * size;
  uint8_t *dst = out + p * lay.stride + lay.offset[i];
  if (size == 4 && ((from | (p * lay.stride + lay.offset[i])) & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(dec) | reinterpret_cast<uintptr_t>(out)) & 3) == 0) {
    *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(dec + from);
  } else {
    for (int b = 0; b < size; b++) dst[b] = dec[from + b];
  }
}

// liblzf stream: ctrl < 32: literal run of ctrl + 1 bytes; else back reference of length
// (ctrl >> 5) + 2 (+ next byte when the 3-bit length is 7) at distance ((ctrl & 31) << 8 | next) + 1.
static pcgx_status lzf_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, size_t *written) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      ctrl++;
      if (op + ctrl > out_len) return fail(PCGX_E_CORRUPT, "lzf: output buffer too small");
      if (ip + ctrl > in_len) return fail(PCGX_E_CORRUPT, "lzf: data corruption");
      memcpy(out + op, in + ip, ctrl);
      ip += ctrl;
      op += ctrl;
    } else {
      size_t len = ctrl >> 5;
      int64_t ref = (int64_t)op - (int64_t)((ctrl & 0x1f) << 8) - 1;
      if (ip >= in_len) return fail(PCGX_E_CORRUPT, "lzf: data corruption");
      if (len == 7) {
        len += in[ip++];
        if (ip >= in_len) return fail(PCGX_E_CORRUPT, "lzf: data corruption");
      }
      ref -= in[ip++];
      len += 2;
      if (op + len > out_len) return fail(PCGX_E_CORRUPT, "lzf: output buffer too small");
      if (ref < 0) return fail(PCGX_E_CORRUPT, "lzf: data corruption");
      for (size_t k = 0; k < len; k++) out[op++] = out[ref++];  // byte by byte: ranges may overlap
    }
  }
  *written = op;
  return PCGX_OK;
}

// ---- Go's strconv on header / ascii tokens
static pcgx_status go_atoi(const std::string &s, int64_t *out) {
  size_t i = 0;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) i++;
  if (i >= s.size()) return fail(PCGX_E_SYNTAX, "strconv.Atoi: parsing \"%s\": invalid syntax", s.c_str());
  for (size_t k = i; k < s.size(); k++)
    if (s[k] < '0' || s[k] > '9') return fail(PCGX_E_SYNTAX, "strconv.Atoi: parsing \"%s\": invalid syntax", s.c_str());
  errno = 0;
  const long long v = strtoll(s.c_str(), nullptr, 10);
  if (errno == ERANGE) return fail(PCGX_E_SYNTAX, "strconv.Atoi: parsing \"%s\": value out of range", s.c_str());
  *out = v;
  return PCGX_OK;
}

static pcgx_status go_parse_float32(const std::string &s, float *out) {
  if (s.empty() || s.find('_') != std::string::npos || s[0] == ' ')
    return fail(PCGX_E_SYNTAX, "strconv.ParseFloat: parsing \"%s\": invalid syntax", s.c_str());
  errno = 0;
  char *end = nullptr;
  const float v = strtof(s.c_str(), &end);  // correctly rounded, like ParseFloat(s, 32)
  if (end != s.c_str() + s.size())
    return fail(PCGX_E_SYNTAX, "strconv.ParseFloat: parsing \"%s\": invalid syntax", s.c_str());
  *out = v;
  return PCGX_OK;
}

static pcgx_status go_parse_uint32(const std::string &s, uint32_t *out) {
  if (s.empty()) return fail(PCGX_E_SYNTAX, "strconv.ParseUint: parsing \"\": invalid syntax");
  uint64_t v = 0;
  for (char c : s) {
    if (c < '0' || c > '9') return fail(PCGX_E_SYNTAX, "strconv.ParseUint: parsing \"%s\": invalid syntax", s.c_str());
    v = v * 10 + (uint64_t)(c - '0');
    if (v > 0xffffffffull) return fail(PCGX_E_SYNTAX, "strconv.ParseUint: parsing \"%s\": value out of range", s.c_str());
  }
  *out = (uint32_t)v;
  return PCGX_OK;
}

// bufio.Reader.ReadLine: the next line without its "\n" / "\r\n"; false at EOF
static bool read_line(const uint8_t *buf, size_t len, size_t *pos, std::string *line) {
  if (*pos >= len) return false;
  const uint8_t *nl = (const uint8_t *)memchr(buf + *pos, '\n', len - *pos);
  const size_t end = nl ? (size_t)(nl - buf) : len;
  size_t e = end;
  if (nl && e > *pos && buf[e - 1] == '\r') e--;
  line->assign((const char *)buf + *pos, e - *pos);
  *pos = nl ? end + 1 : len;
  return true;
}

static std::vector<std::string> go_fields(const std::string &s) {  // strings.Fields
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\v' || s[i] == '\f')) i++;
    size_t j = i;
    while (j < s.size() && !(s[j] == ' ' || s[j] == '\t' || s[j] == '\r' || s[j] == '\v' || s[j] == '\f')) j++;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
  return out;
}

static PcdLayout make_layout(const pcgx_pcd_header *h) {
  PcdLayout lay;
  memset(&lay, 0, sizeof lay);
  lay.n_fields = h->n_fields;
  int64_t pos = 0;
  int32_t off = 0;
  for (int i = 0; i < h->n_fields; i++) {  // io.go:208-215
    lay.size[i] = h->size[i];
    lay.head[i] = pos;
    lay.offset[i] = off;
    pos += (int64_t)h->size[i] * h->count[i] * h->points;
    off += h->size[i] * h->count[i];
  }
  lay.stride = h->stride;
  return lay;
}

// POINTS x stride, and every field block's end, without int64 wrap-around: a header is untrusted
// input, and every bounds check below is made in terms of these products.
static pcgx_status payload_bytes(const pcgx_pcd_header *h, int64_t *total) {
  if (h->n_fields < 0 || h->n_fields > PCGX_PCD_MAX_FIELDS || h->points < 0 || h->stride < 0)
    return fail(PCGX_E_BAD_HEADER, "bad header (fields / POINTS / stride out of range)");
  int64_t stride = 0;
  for (int i = 0; i < h->n_fields; i++) {
    if (h->size[i] < 0 || h->count[i] < 0) return fail(PCGX_E_BAD_HEADER, "negative SIZE / COUNT");
    stride += (int64_t)h->size[i] * h->count[i];
  }
  if (stride != h->stride) return fail(PCGX_E_BAD_HEADER, "stride does not match SIZE x COUNT");
  if (__builtin_mul_overflow(h->points, h->stride, total) || *total > ((int64_t)1 << 60))
    return fail(PCGX_E_BAD_HEADER, "POINTS x stride overflows");
  return PCGX_OK;
}

// Payload of a binary_compressed file, LZF-decoded (field-major blocks).
static pcgx_status decode_compressed(const uint8_t *file, size_t len, const pcgx_pcd_header *h,
                                     std::vector<uint8_t> *dec) {
  size_t pos = (size_t)h->data_offset;
  if (len - pos < 4) return fail(PCGX_E_EOF, "EOF");  // binary.Read(&nCompressed), io.go:186-188
  int32_t ncomp, nunc;
  memcpy(&ncomp, file + pos, 4);
  pos += 4;
  if (len - pos < 4) return fail(PCGX_E_EOF, "EOF");
  memcpy(&nunc, file + pos, 4);
  pos += 4;
  if (ncomp < 0 || nunc < 0) return fail(PCGX_E_BAD_HEADER, "negative compressed / uncompressed size");
  if (len - pos < (size_t)ncomp) return fail(PCGX_E_EOF, "EOF");  // io.ReadFull, io.go:194-196
  dec->assign((size_t)nunc, 0);
  size_t got = 0;
  PCGX_TRY(lzf_decompress(file + pos, (size_t)ncomp, dec->data(), dec->size(), &got));
  if (got != (size_t)nunc) return fail(PCGX_E_BAD_HEADER, "wrong uncompressed size");  // io.go:201-203
  // the reference would panic on a stream shorter than the header promises: an error here
  {
    int64_t t0;
    PCGX_TRY(payload_bytes(h, &t0));  // from here on size x count x POINTS cannot wrap (each <= POINTS x stride)
  }
  const PcdLayout lay = make_layout(h);
  for (int i = 0; i < lay.n_fields; i++)
    if (h->points > 0 && lay.head[i] + (h->points - 1) * (int64_t)lay.size[i] + lay.size[i] > (int64_t)nunc)
      return fail(PCGX_E_OUT_OF_RANGE, "binary_compressed payload shorter than POINTS x fields (the reference panics)");
  int64_t total;
  PCGX_TRY(payload_bytes(h, &total));
  if ((int64_t)nunc < total && h->points > 0)
    return fail(PCGX_E_OUT_OF_RANGE, "binary_compressed payload shorter than POINTS x stride (the reference panics)");
  return PCGX_OK;
}

static pcgx_status parse_ascii(const uint8_t *file, size_t len, const pcgx_pcd_header *h, uint8_t *out) {
  int64_t total;
  PCGX_TRY(payload_bytes(h, &total));
  memset(out, 0, (size_t)total);
  size_t pos = (size_t)h->data_offset;
  int64_t data_off = 0;
  std::string line;
  while (read_line(file, len, &pos, &line)) {  // io.go:143-180
    const std::vector<std::string> tok = go_fields(line);
    size_t line_off = 0;
    for (int i = 0; i < h->n_fields; i++) {
      for (int j = 0; j < h->count[i]; j++) {
        if (h->type[i] == 'F' || h->type[i] == 'U') {
          if (line_off + j >= tok.size() || data_off + 4 > total)
            return fail(PCGX_E_OUT_OF_RANGE, "ascii payload does not match the header (the reference panics: index out of range)");
          uint32_t bits;
          if (h->type[i] == 'F') {
            float v;
            PCGX_TRY(go_parse_float32(tok[line_off + j], &v));
            memcpy(&bits, &v, 4);
          } else {
            PCGX_TRY(go_parse_uint32(tok[line_off + j], &bits));
          }
          memcpy(out + data_off, &bits, 4);  // binary.LittleEndian.PutUint32
        }
        data_off += h->size[i];
      }
      line_off += (size_t)h->count[i];
    }
  }
  return PCGX_OK;
}

}  // namespace pcgx

using namespace pcgx;

extern "C" pcgx_status pcgx_pcd_unmarshal_header(const void *file, size_t len, pcgx_pcd_header *h) {
  PCGX_API_LOCK();
  if (!h || (len > 0 && !file)) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal_header: NULL argument");
  memset(h, 0, sizeof *h);
  const uint8_t *buf = (const uint8_t *)file;
  size_t pos = 0;
  std::string line;
  int n_size = -1, n_type = -1, n_count = -1, n_fields = 0;
  bool have_data = false;
  while (!have_data) {
    if (!read_line(buf, len, &pos, &line)) return fail(PCGX_E_EOF, "EOF");  // io.go:52-55
    const std::vector<std::string> a = go_fields(line);
    if (a.size() < 2) return fail(PCGX_E_BAD_HEADER, "header field must have value");
    const size_t nv = a.size() - 1;
    const std::string &k = a[0];
    if (k == "FIELDS" || k == "SIZE" || k == "TYPE" || k == "COUNT") {
      if (nv > PCGX_PCD_MAX_FIELDS) return fail(PCGX_E_TOO_LARGE, "more than %d fields", PCGX_PCD_MAX_FIELDS);
    }
    if (k == "VERSION") {
      PCGX_TRY(go_parse_float32(a[1], &h->version));
    } else if (k == "FIELDS") {
      n_fields = (int)nv;
      for (size_t i = 0; i < nv; i++) {
        if (a[i + 1].size() >= sizeof h->fields[0]) return fail(PCGX_E_TOO_LARGE, "field name longer than 31 bytes");
        memset(h->fields[i], 0, sizeof h->fields[i]);
        memcpy(h->fields[i], a[i + 1].data(), a[i + 1].size());
      }
    } else if (k == "SIZE") {
      n_size = (int)nv;
      for (size_t i = 0; i < nv; i++) {
        int64_t v;
        PCGX_TRY(go_atoi(a[i + 1], &v));
        if (v > INT32_MAX || v < INT32_MIN) return fail(PCGX_E_BAD_HEADER, "SIZE does not fit 32 bits");
        h->size[i] = (int32_t)v;
      }
    } else if (k == "TYPE") {
      n_type = (int)nv;
      for (size_t i = 0; i < nv; i++) h->type[i] = a[i + 1].size() == 1 ? a[i + 1][0] : '?';
    } else if (k == "COUNT") {
      n_count = (int)nv;
      for (size_t i = 0; i < nv; i++) {
        int64_t v;
        PCGX_TRY(go_atoi(a[i + 1], &v));
        if (v > INT32_MAX || v < INT32_MIN) return fail(PCGX_E_BAD_HEADER, "COUNT does not fit 32 bits");
        h->count[i] = (int32_t)v;
      }
    } else if (k == "WIDTH") {
      PCGX_TRY(go_atoi(a[1], &h->width));
    } else if (k == "HEIGHT") {
      PCGX_TRY(go_atoi(a[1], &h->height));
    } else if (k == "VIEWPOINT") {
      if (nv > 16) return fail(PCGX_E_TOO_LARGE, "more than 16 viewpoint values");
      h->n_viewpoint = (int32_t)nv;
      for (size_t i = 0; i < nv; i++) PCGX_TRY(go_parse_float32(a[i + 1], &h->viewpoint[i]));
    } else if (k == "POINTS") {
      PCGX_TRY(go_atoi(a[1], &h->points));
    } else if (k == "DATA") {
      if (a[1] == "ascii") h->format = PCGX_PCD_ASCII;
      else if (a[1] == "binary") h->format = PCGX_PCD_BINARY;
      else if (a[1] == "binary_compressed") h->format = PCGX_PCD_BINARY_COMPRESSED;
      else return fail(PCGX_E_BAD_HEADER, "unknown data format");
      have_data = true;
    }
  }
  h->n_fields = n_fields;
  // validate (io.go:125-134); a missing line counts as length 0
  if (n_fields != (n_size < 0 ? 0 : n_size)) return fail(PCGX_E_BAD_HEADER, "size field size is wrong");
  if (n_fields != (n_type < 0 ? 0 : n_type)) return fail(PCGX_E_BAD_HEADER, "type field size is wrong");
  if (n_fields != (n_count < 0 ? 0 : n_count)) return fail(PCGX_E_BAD_HEADER, "count field size is wrong");
  int64_t stride = 0;
  for (int i = 0; i < n_fields; i++) {
    if (h->size[i] < 0 || h->count[i] < 0) return fail(PCGX_E_BAD_HEADER, "negative SIZE / COUNT");
    stride += (int64_t)h->size[i] * h->count[i];  // pointcloud.go:64-70 (<= 64 x 2^62: no overflow)
  }
  if (h->points < 0) return fail(PCGX_E_BAD_HEADER, "negative POINTS (the reference panics in make)");
  h->stride = stride;
  h->data_offset = (int64_t)pos;
  int64_t total;
  return payload_bytes(h, &total);  // POINTS x stride must be a size a buffer can have
}

extern "C" pcgx_status pcgx_pcd_unmarshal(const void *file, size_t len, const pcgx_pcd_header *h, void *out_data) {
  PCGX_API_LOCK();
  if (!h || (len > 0 && !file)) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal: NULL argument");
  int64_t total;
  PCGX_TRY(payload_bytes(h, &total));
  if (total > 0 && !out_data) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal: out_data is NULL");
  if (h->data_offset < 0 || (size_t)h->data_offset > len) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal: bad header");
  const uint8_t *buf = (const uint8_t *)file;
  uint8_t *out = (uint8_t *)out_data;
  if (h->format == PCGX_PCD_ASCII) return parse_ascii(buf, len, h, out);
  if (h->format == PCGX_PCD_BINARY) {
    // io.ReadFull (io.go:181-185): EOF when nothing is left, ErrUnexpectedEOF when too little
    if ((int64_t)(len - (size_t)h->data_offset) < total) return fail(PCGX_E_EOF, "EOF");
    if (total > 0) memcpy(out, buf + h->data_offset, (size_t)total);
    return PCGX_OK;
  }
  std::vector<uint8_t> dec;
  PCGX_TRY(decode_compressed(buf, len, h, &dec));
  const PcdLayout lay = make_layout(h);
  memset(out, 0, (size_t)total);  // pp.Data = make([]byte, n) (io.go:218)
  for (int64_t p = 0; p < h->points; p++)
    for (int i = 0; i < lay.n_fields; i++)
      memcpy(out + p * lay.stride + lay.offset[i], dec.data() + lay.head[i] + p * lay.size[i], (size_t)lay.size[i]);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_pcd_unmarshal_dev(const void *file, size_t len, const pcgx_pcd_header *h, void *d_out,
                                              void *stream) {
  PCGX_API_LOCK();
  if (!h || (len > 0 && !file)) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal_dev: NULL argument");
  int64_t total;
  PCGX_TRY(payload_bytes(h, &total));
  if (total > 0 && !d_out) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal_dev: d_out is NULL");
  if (h->data_offset < 0 || (size_t)h->data_offset > len) return fail(PCGX_E_INVALID, "pcgx_pcd_unmarshal_dev: bad header");
  if (total == 0) return PCGX_OK;
  PCGX_TRY(ensure_init());
  hipStream_t st = pick_stream(stream);
  const uint8_t *buf = (const uint8_t *)file;
  if (h->format == PCGX_PCD_BINARY) {  // the payload already is the AoS record array
    if ((int64_t)(len - (size_t)h->data_offset) < total) return fail(PCGX_E_EOF, "EOF");
    PCGX_HIP_TRY(hipMemcpyAsync(d_out, buf + h->data_offset, (size_t)total, hipMemcpyHostToDevice, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));  // the caller's buffer may go away
    return PCGX_OK;
  }
  if (h->format == PCGX_PCD_ASCII) {
    std::vector<uint8_t> rec((size_t)total);
    PCGX_TRY(parse_ascii(buf, len, h, rec.data()));
    PCGX_HIP_TRY(hipMemcpyAsync(d_out, rec.data(), (size_t)total, hipMemcpyHostToDevice, st));
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    return PCGX_OK;
  }
  std::vector<uint8_t> dec;
  PCGX_TRY(decode_compressed(buf, len, h, &dec));
  Arena &ar = ctx().arena;
  PCGX_TRY(ar.begin(st));
  uint8_t *d_dec = nullptr;
  PCGX_TRY(ar.alloc_n(dec.size(), &d_dec));
  PCGX_HIP_TRY(hipMemcpyAsync(d_dec, dec.data(), dec.size(), hipMemcpyHostToDevice, st));
  PCGX_HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)total, st));
  const PcdLayout lay = make_layout(h);
  const int64_t work = h->points * lay.n_fields;
  hipLaunchKernelGGL(pcd_deinterleave_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st,
                     (const uint8_t *)d_dec, (int64_t)dec.size(), h->points, lay, (uint8_t *)d_out);
  PCGX_HIP_TRY(hipGetLastError());
  PCGX_HIP_TRY(hipStreamSynchronize(st));  // `dec` is a host temporary
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_pcd_marshal(const pcgx_pcd_header *h, const void *data, void *out, size_t cap,
                                        size_t *out_len) {
  PCGX_API_LOCK();
  if (!h || !out_len) return fail(PCGX_E_INVALID, "pcgx_pcd_marshal: NULL argument");
  const int64_t total = h->points * h->stride;
  if (total > 0 && !data) return fail(PCGX_E_INVALID, "pcgx_pcd_marshal: data is NULL");
  auto join_int = [&](const int32_t *v) {
    std::string s;
    for (int i = 0; i < h->n_fields; i++) s += (i ? " " : "") + std::to_string(v[i]);
    return s;
  };
  std::string fields, types, vp;
  for (int i = 0; i < h->n_fields; i++) {
    fields += (i ? " " : "") + std::string(h->fields[i]);
    types += (i ? " " : "") + std::string(1, h->type[i]);
  }
  // a missing viewpoint gets the default pcl_viewer needs (io.go:248-254)
  const float def_vp[7] = {0, 0, 0, 1, 0, 0, 0};
  const int nvp = h->n_viewpoint == 0 ? 7 : h->n_viewpoint;
  const float *vpv = h->n_viewpoint == 0 ? def_vp : h->viewpoint;
  char num[64];
  for (int i = 0; i < nvp; i++) {
    snprintf(num, sizeof num, "%.4f", (double)vpv[i]);  // FormatFloat(float64(v), 'f', 4, 32)
    vp += (i ? " " : "") + std::string(num);
  }
  snprintf(num, sizeof num, "%0.1f", (double)h->version);
  std::string head = "VERSION " + std::string(num) + "\nFIELDS " + fields + "\nSIZE " + join_int(h->size) + "\nTYPE " +
                     types + "\nCOUNT " + join_int(h->count) + "\nWIDTH " + std::to_string(h->width) + "\nHEIGHT " +
                     std::to_string(h->height) + "\nVIEWPOINT " + vp + "\nPOINTS " + std::to_string(h->points) +
                     "\nDATA binary\n";
  *out_len = head.size() + (size_t)total;
  if (!out || cap < *out_len) return out ? fail(PCGX_E_INVALID, "pcgx_pcd_marshal: buffer too small") : PCGX_OK;
  memcpy(out, head.data(), head.size());
  if (total > 0) memcpy((uint8_t *)out + head.size(), data, (size_t)total);
  return PCGX_OK;
}
