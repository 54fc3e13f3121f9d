// plink_io.cpp -- see plink_io.hpp.  Behaviour (line handling, error texts, number format) follows the reference's
// data.cpp:419-672 and util.h:69-108 so that files round-trip between the two programs.
#include "plink_io.hpp"

#include <cerrno>
#include <cstdlib>
#include <charconv>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <algorithm>
#include <cstdio>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace fpca {

namespace {

// every '\n'-terminated line (the reference's `if(!in.eof())` after getline drops an unterminated tail)
std::vector<std::string> terminated_lines(std::ifstream &in)
{
   std::vector<std::string> lines;
   std::string line;
   while (std::getline(in, line)) {
      if (in.eof()) break; // last chunk had no trailing newline
      lines.push_back(line);
   }
   return lines;
}

// whitespace-separated fields of one line, as `stream >> token` would extract them (a stringstream per line costs more
// than parsing the numbers: 0.27 s for the .fam of 500,000 samples); the views point into `s`
struct Token {
   const char *p;
   size_t n;
   std::string str() const { return std::string(p, n); }
   bool operator!=(const std::string &o) const { return o.size() != n || std::memcmp(o.data(), p, n) != 0; }
};
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }
void split_ws(const std::string &s, std::vector<Token> &tok)
{
   tok.clear();
   const char *p = s.data(), *e = p + s.size();
   while (p < e) {
      while (p < e && is_ws(*p)) p++;
      if (p == e) break;
      const char *q = p;
      while (q < e && !is_ws(*q)) q++;
      tok.push_back(Token{p, (size_t)(q - p)});
      p = q;
   }
}
// strtod / strtol over a token: the character after it is whitespace or the string's terminating NUL, so the C parsers
// stop there by themselves; "fully parsed" = they stopped exactly at the token's end.  This IS the reference's rule
// (data.cpp:571-579: strtod, then `*err != '\0' || errno != 0` is an error) -- so "nan" / "inf" / hex floats are accepted
// and a subnormal that raises ERANGE is rejected there as here; operator>>(double) is not what the reference uses.
inline bool parse_double(const Token &t, double &v)
{
   char *end = nullptr;
   errno = 0;
   v = std::strtod(t.p, &end);
   return end == t.p + t.n && errno == 0;
}
inline bool parse_long(const Token &t)
{
   char *end = nullptr;
   errno = 0;
   (void)std::strtol(t.p, &end, 10);
   return end == t.p + t.n && errno == 0;
}

} // namespace

TextMatrix read_text(const std::string &filename, unsigned firstcol, long nrows, unsigned skip)
{
   std::ifstream in(filename, std::ios::in);
   if (!in) throw std::runtime_error("Error reading file '" + filename + "': " + strerror(errno));
   std::vector<std::string> all = terminated_lines(in), lines;
   unsigned line_num = 0;
   for (auto &l : all) {
      if (nrows == -1 || (long)line_num < nrows) {
         if (line_num >= skip) lines.push_back(l);
         line_num++;
      }
   }
   TextMatrix M;
   uint64_t numfields_1st = 0;
   std::vector<Token> tokens;
   for (size_t i = 0; i < lines.size(); i++) {
      split_ws(lines[i], tokens);
      if (tokens.size() + 1 < firstcol + 1u && tokens.size() < firstcol)
         throw std::runtime_error("Error reading file '" + filename + "': inconsistent number of columns");
      const uint64_t numfields = tokens.size() - firstcol + 1;
      if (i == 0) {
         M.rows = lines.size();
         M.cols = numfields;
         M.v.assign(M.rows * M.cols, 0.0);
         numfields_1st = numfields;
      } else if (numfields != numfields_1st)
         throw std::runtime_error("Error reading file '" + filename + "': inconsistent number of columns");
      for (uint64_t j = 0; j < numfields; j++) {
         const Token &t = tokens[j + firstcol - 1];
         double m;
         if (!parse_double(t, m))
            throw std::runtime_error("Error reading file '" + filename + "', line " + std::to_string(i + 1) + ": '" + t.str() +
                                     "' cannot be parsed as a number");
         M.at(i, j) = m;
      }
   }
   return M;
}

void read_plink_fam(const std::string &filename, std::vector<std::string> &fam_ids, std::vector<std::string> &indiv_ids)
{
   std::ifstream in(filename, std::ios::in);
   if (!in) throw std::runtime_error("[Data::read_plink_fam] Error reading file " + filename);
   std::vector<Token> tokens;
   const std::vector<std::string> lines = terminated_lines(in);
   fam_ids.reserve(fam_ids.size() + lines.size());
   indiv_ids.reserve(indiv_ids.size() + lines.size());
   for (auto &l : lines) {
      split_ws(l, tokens);
      if (tokens.size() < 2) throw std::runtime_error("[Data::read_plink_fam] malformed line in " + filename);
      fam_ids.push_back(tokens[0].str());
      indiv_ids.push_back(tokens[1].str());
   }
}

void read_plink_bim(const std::string &filename, std::vector<std::string> &snp_ids, std::vector<std::string> &ref_alleles,
                    std::vector<std::string> &alt_alleles)
{
   std::ifstream in(filename, std::ios::in);
   if (!in) throw std::runtime_error("Error reading file " + filename);
   std::vector<std::string> lines = terminated_lines(in);
   std::vector<Token> tokens;
   for (size_t i = 0; i < lines.size(); i++) {
      split_ws(lines[i], tokens);
      if (tokens.size() < 6) throw std::runtime_error("Error reading file '" + filename + "', line " + std::to_string(i + 1) + ": too few columns");
      snp_ids.push_back(tokens[1].str());
      ref_alleles.push_back(tokens[4].str());
      alt_alleles.push_back(tokens[5].str());
      if (!parse_long(tokens[3]))
         throw std::runtime_error("Error reading file '" + filename + "', line " + std::to_string(i + 1) + ": '" + tokens[3].str() +
                                  "' cannot be parsed as a number");
   }
}

std::vector<double> read_maf(const std::string &filename, const std::vector<std::string> &snp_ids)
{
   std::ifstream in(filename, std::ios::in);
   if (!in) throw std::runtime_error("Error reading file '" + filename + "': " + strerror(errno));
   std::vector<std::string> lines = terminated_lines(in);
   if (!lines.empty()) lines.erase(lines.begin()); // header of the .frq
   if (lines.size() != snp_ids.size())
      throw std::runtime_error("Error number of SNPs in '" + filename + "': different number of SNPs than in the bim file'");
   std::vector<double> maf(lines.size());
   std::vector<Token> tokens;
   for (size_t i = 0; i < lines.size(); i++) {
      split_ws(lines[i], tokens);
      if (tokens.size() != 6) throw std::runtime_error("Error reading file '" + filename + "': inconsistent number of columns");
      if (tokens[1] != snp_ids[i])
         throw std::runtime_error("Error reading file '" + filename + "': inconsistent SNP id at row':" + std::to_string(i));
      if (!parse_double(tokens[4], maf[i]))
         throw std::runtime_error("Error reading file '" + filename + "', line " + std::to_string(i + 1) + ": '" + tokens[4].str() +
                                  "' cannot be parsed as a number");
   }
   return maf;
}

namespace {

// "%.{p}g" is what operator<<(double) under std::setprecision(p) (default floatfield) produces: num_put formats
// through the printf conversion %g with the stream precision ([facet.num.put.virtuals]); util.h:77,97 rely on it.
// std::to_chars(general, precision) is specified to give the same characters as printf's %.{p}g in the "C" locale and
// is several times faster (no locale, no format parsing); non-finite values keep the printf route.
inline void append_number(std::string &out, double v, unsigned precision)
{
   char buf[64];
   if (std::isfinite(v)) {
      const std::to_chars_result r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::general, (int)precision);
      if (r.ec == std::errc()) {
         out.append(buf, (size_t)(r.ptr - buf));
         return;
      }
   }
   const int n = std::snprintf(buf, sizeof(buf), "%.*g", (int)precision, v);
   out.append(buf, (size_t)n);
}

void format_rows(const double *M, uint64_t rows, uint64_t cols, const std::vector<std::string> &rownames, uint64_t r0,
                 uint64_t r1, unsigned precision, std::string &out)
{
   out.clear();
   out.reserve((size_t)(r1 - r0) * (cols * 14 + 24));
   for (uint64_t j = r0; j < r1; j++) {
      if (!rownames.empty()) {
         out += rownames[j];
         out += '\t';
      }
      for (uint64_t c = 0; c < cols; c++) {
         if (c) out += '\t';
         append_number(out, M[j + c * rows], precision);
      }
      out += '\n';
   }
}

} // namespace

// The eigenvector / PC files are N x k numbers of text (110 MB at N = 200,000, k = 20): rows are formatted in parallel
// chunks and written in order, so the bytes are identical to the reference's serial operator<< loop.
bool save_text(const double *M, uint64_t rows, uint64_t cols, const std::vector<std::string> &colnames,
               const std::vector<std::string> &rownames, const std::string &filename, unsigned precision)
{
   std::ofstream out(filename, std::ofstream::out | std::ofstream::binary);
   if (!out) {
      std::cerr << "Error while saving to file " << filename << ":" << strerror(errno) << std::endl;
      return false;
   }
   std::string header;
   for (size_t i = 0; i < colnames.size(); i++) {
      header += colnames[i];
      header += (i == colnames.size() - 1) ? "\n" : "\t";
   }
   out.write(header.data(), (std::streamsize)header.size());
   const uint64_t chunk = 4096;
   const uint64_t nchunks = (rows + chunk - 1) / chunk;
   unsigned nthreads = std::thread::hardware_concurrency();
   if (nthreads == 0) nthreads = 1;
   if (nthreads > 32) nthreads = 32;
   if ((uint64_t)nthreads > nchunks) nthreads = (unsigned)(nchunks ? nchunks : 1);
   // waves of nthreads chunks: format concurrently, then write the wave in order
   std::vector<std::string> bufs(nthreads);
   for (uint64_t c0 = 0; c0 < nchunks; c0 += nthreads) {
      const unsigned nw = (unsigned)std::min<uint64_t>(nthreads, nchunks - c0);
      std::vector<std::thread> th;
      for (unsigned t = 1; t < nw; t++)
         th.emplace_back([&, t] {
            const uint64_t r0 = (c0 + t) * chunk;
            format_rows(M, rows, cols, rownames, r0, std::min(rows, r0 + chunk), precision, bufs[t]);
         });
      format_rows(M, rows, cols, rownames, c0 * chunk, std::min(rows, c0 * chunk + chunk), precision, bufs[0]);
      for (auto &t : th) t.join();
      for (unsigned t = 0; t < nw; t++) out.write(bufs[t].data(), (std::streamsize)bufs[t].size());
   }
   out.close();
   return (bool)out;
}

} // namespace fpca
