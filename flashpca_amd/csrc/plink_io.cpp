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
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <sched.h>

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

uint64_t read_fam(const std::string &filename, std::vector<std::string> &fam_ids, std::vector<std::string> &indiv_ids)
{
   // the whole file in one read (13 MB at 500,000 samples), lines and fields as views into it: one pass instead of the two
   // getline passes of read_text + read_plink_fam (103 -> ~35 ms at that size)
   std::ifstream in(filename, std::ios::in | std::ios::binary);
   if (!in) throw std::runtime_error("Error reading file '" + filename + "': " + strerror(errno));
   std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
   uint64_t nlines = 0;
   for (char c : all) nlines += c == '\n'; // '\n'-terminated lines only (data.cpp:526: an unterminated tail is dropped)
   fam_ids.reserve(fam_ids.size() + nlines);
   indiv_ids.reserve(indiv_ids.size() + nlines);
   std::vector<Token> tokens;
   uint64_t numfields_1st = 0, i = 0;
   const char *p = all.data(), *e = p + all.size();
   bool malformed_ids = false;
   while (p < e) {
      const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
      if (!nl) break;
      // fields of [p, nl): the byte at nl is whitespace, so strtod stops at a field's end by itself
      tokens.clear();
      for (const char *q = p; q < nl;) {
         while (q < nl && is_ws(*q)) q++;
         if (q == nl) break;
         const char *t = q;
         while (q < nl && !is_ws(*q)) q++;
         tokens.push_back(Token{t, (size_t)(q - t)});
      }
      // read_text(filename, 6): data.cpp:548-583
      if (tokens.size() < 6) throw std::runtime_error("Error reading file '" + filename + "': inconsistent number of columns");
      const uint64_t numfields = tokens.size() - 5;
      if (i == 0)
         numfields_1st = numfields;
      else if (numfields != numfields_1st)
         throw std::runtime_error("Error reading file '" + filename + "': inconsistent number of columns");
      for (uint64_t j = 0; j < numfields; j++) {
         double m;
         if (!parse_double(tokens[j + 5], m))
            throw std::runtime_error("Error reading file '" + filename + "', line " + std::to_string(i + 1) + ": '" + tokens[j + 5].str() +
                                     "' cannot be parsed as a number");
      }
      if (tokens.size() < 2) malformed_ids = true; // (cannot happen after the six-field check; kept for read_plink_fam's rule)
      fam_ids.emplace_back(tokens[0].p, tokens[0].n);
      indiv_ids.emplace_back(tokens[1].p, tokens[1].n);
      i++;
      p = nl + 1;
   }
   if (malformed_ids) throw std::runtime_error("[Data::read_plink_fam] malformed line in " + filename);
   return i;
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
// rows [r0, r1) as text.  The characters go straight into a buffer sized for the worst case (a %.{p}g number is at most
// p + 8 characters: sign, point, "e-308"; 32 covers every precision the CLI accepts up to 24) -- no per-number temporary, no
// capacity checks: 138 -> ~100 ns per number and thread, and the writer threads are what the output phase consists of.
void format_rows(const double *M, uint64_t rows, uint64_t cols, const std::vector<std::string> &rownames, uint64_t r0,
                 uint64_t r1, unsigned precision, std::string &out)
{
   // (a double has at most 767 significant decimal digits -- the smallest denormal -- so %.{p}g never writes more than
   //  min(p, 767) + 8 characters whatever --precision says)
   const size_t numw = (size_t)std::max(32u, std::min(precision, 767u) + 10u);
   size_t bound = 0;
   for (uint64_t j = r0; j < r1; j++) bound += (rownames.empty() ? 0 : rownames[j].size() + 1) + cols * (numw + 1) + 1;
   out.resize(bound);
   char *p = out.data();
   for (uint64_t j = r0; j < r1; j++) {
      if (!rownames.empty()) {
         std::memcpy(p, rownames[j].data(), rownames[j].size());
         p += rownames[j].size();
         *p++ = '\t';
      }
      for (uint64_t c = 0; c < cols; c++) {
         if (c) *p++ = '\t';
         const double v = M[j + c * rows];
         bool done = false;
         if (std::isfinite(v)) {
            const std::to_chars_result r = std::to_chars(p, p + numw, v, std::chars_format::general, (int)precision);
            if (r.ec == std::errc()) {
               p = r.ptr;
               done = true;
            }
         }
         if (!done) p += std::snprintf(p, numw, "%.*g", (int)precision, v);
      }
      *p++ = '\n';
   }
   out.resize((size_t)(p - out.data()));
}

} // namespace

// The eigenvector / PC files are N x k numbers of text (140 MB each at N = 500,000, k = 20).  Rows are formatted in chunks
// by a pool of worker threads and written IN ORDER by the calling thread while the workers format the chunks behind -- a
// ring of 2 T chunk buffers -- so the bytes are identical to the reference's serial operator<< loop, and neither the
// formatting nor the write() calls wait for the other (round 2 formatted a wave of chunks, then wrote it, then formatted the
// next: 337 ms of the 1.25 s the CLI took at the headline size).
bool save_text(const double *M, uint64_t rows, uint64_t cols, const std::vector<std::string> &colnames,
               const std::vector<std::string> &rownames, const std::string &filename, unsigned precision, unsigned max_threads)
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
   // rows per chunk: 4096 at ordinary precisions; fewer when --precision asks for numbers hundreds of characters wide, so that
   // the 2 T chunk buffers (worst-case sized, see format_rows) stay around 8 MB each whatever the precision
   const uint64_t numw = std::max<uint64_t>(32, std::min<uint64_t>(precision, 767) + 10);
   const uint64_t chunk = std::max<uint64_t>(16, std::min<uint64_t>(4096, (8ull << 20) / std::max<uint64_t>(1, cols * (numw + 1))));
   const uint64_t nchunks = (rows + chunk - 1) / chunk;
   uint64_t T = max_threads ? max_threads : usable_cpus();
   if (T > 32) T = 32;
   if (T > nchunks) T = nchunks;
   if (T <= 1) { // small files: one thread, no pool
      std::string buf;
      for (uint64_t c = 0; c < nchunks; c++) {
         format_rows(M, rows, cols, rownames, c * chunk, std::min(rows, c * chunk + chunk), precision, buf);
         out.write(buf.data(), (std::streamsize)buf.size());
      }
      out.close();
      return (bool)out;
   }
   const uint64_t R = 2 * T;
   std::vector<std::string> bufs(R);
   std::vector<uint64_t> ready(R, 0); // ready[slot] == i + 1: chunk i is formatted in bufs[slot]   (all under `mu`)
   uint64_t next = 0, written = 0;
   std::mutex mu;
   std::condition_variable cv_ready, cv_free; // no spinning: the CPUs belong to the formatters (and to the other files' writers)
   std::vector<std::thread> workers;
   for (uint64_t t = 0; t < T; t++)
      workers.emplace_back([&] {
         for (;;) {
            uint64_t i;
            {
               std::unique_lock<std::mutex> lk(mu);
               i = next++;
               if (i >= nchunks) return;
               cv_free.wait(lk, [&] { return written + R > i; }); // chunk i - R has left its slot
            }
            format_rows(M, rows, cols, rownames, i * chunk, std::min(rows, i * chunk + chunk), precision, bufs[i % R]);
            {
               std::lock_guard<std::mutex> lk(mu);
               ready[i % R] = i + 1;
            }
            cv_ready.notify_one();
         }
      });
   for (uint64_t i = 0; i < nchunks; i++) {
      {
         std::unique_lock<std::mutex> lk(mu);
         cv_ready.wait(lk, [&] { return ready[i % R] == i + 1; });
      }
      out.write(bufs[i % R].data(), (std::streamsize)bufs[i % R].size());
      {
         std::lock_guard<std::mutex> lk(mu);
         written = i + 1;
      }
      cv_free.notify_all();
   }
   for (auto &w : workers) w.join();
   out.close();
   return (bool)out;
}

} // namespace fpca
