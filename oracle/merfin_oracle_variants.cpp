/*
 * merfin_oracle_variants.cpp -- CPU restatement of merfin's variant modes
 * (-filter / -polish / -better / -strict / -loose).  TEST INFRASTRUCTURE ONLY,
 * same rules and the same "parity unpinned" status as merfin_oracle.h.
 *
 * C++ (not C) because the reference's observable behaviour depends on
 * libstdc++ semantics that have to be reproduced literally: std::sort on
 * cluster start (unstable, vcf.C:176-178), std::multimap<double,int,
 * greater<int>> as the tie-breaker of bestVariant (varMer.H:72), std::string
 * ::replace in traverse, std::list::sort/unique in bestFilter.
 *
 * Follows, statement by statement (paths relative to /root/reference):
 *   vcfRecord::load/save            src/merfin/vcfRecord.H:50-100
 *   gtAllele::gtAllele              src/merfin/vcf.C:23-87
 *   vcfFile::loadFile               src/merfin/vcf.C:93-149
 *   vcfFile::mergeChrPosGT          src/merfin/vcf.C:156-246
 *   traverse                        src/merfin/merfin-variants.C:22-126
 *   processVariants/outputVariants  src/merfin/merfin-variants.C:131-345
 *   varMer::*                       src/merfin/varMer.C:37-659
 * splitToWords (meryl-utility, absent) is restated from its call sites: runs
 * of separator characters collapse, operator[] past the end yields nullptr.
 */
#include "merfin_oracle.h"

#include <assert.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <list>
#include <map>
#include <string>
#include <vector>

using namespace std;

enum { OP_FILTER = 4, OP_POLISH = 5, OP_BETTER = 6, OP_STRICT = 7, OP_LOOSE = 8 };   /* merfin-globals.H:34-38 */

namespace {

/* splitToWords restated: tokens separated by any char of `seps`, empty tokens dropped */
struct Words {
  string buf;
  vector<char *> w;
  void split(const char *s, const char *seps) {
    buf = s ? s : "";
    w.clear();
    char *p = &buf[0];
    size_t n = buf.size();
    size_t i = 0;
    while (i < n) {
      while (i < n && strchr(seps, p[i])) p[i++] = 0;
      if (i >= n) break;
      w.push_back(p + i);
      while (i < n && !strchr(seps, p[i])) i++;
    }
  }
  size_t numWords() const { return w.size(); }
  char *operator[](size_t i) const { return i < w.size() ? w[i] : nullptr; }
};

struct vcfRecord {                                       /* vcfRecord.H */
  Words _words;
  char *_chr = nullptr; uint32_t _pos = 0xffffffffu; char *_id = nullptr, *_ref = nullptr, *_alts = nullptr;
  double _qual = 0.0; char *_filter = nullptr, *_info = nullptr, *_formats = nullptr, *_samples = nullptr;
  Words _arr_alts, _arr_formats, _arr_samples;
  bool _isValid = false;

  bool load(const char *inLine) {                        /* :50-76 */
    _words.split(inLine, "\t");
    if (_words.numWords() < 10)
      return false;
    _chr = _words[0];
    _pos = (uint32_t)strtoul(_words[1], nullptr, 10);
    _id = _words[2];
    _ref = _words[3];
    _alts = _words[4];
    _qual = strtod(_words[5], nullptr);
    _filter = _words[6];
    _info = _words[7];
    _formats = _words[8];
    _samples = _words[9];
    _arr_alts.split(_alts, ",");
    _arr_formats.split(_formats, ":");
    _arr_samples.split(_samples, ":");
    _isValid = true;
    return true;
  }
  string save() {                                        /* :83-100 */
    char qual[64];
    snprintf(qual, sizeof(qual), "%.1f", _qual);
    return string(_chr) + "\t" + to_string((int)_pos) + "\t" + _id + "\t" + _ref + "\t" + _alts + "\t" + qual + "\t" +
           _filter + "\t" + _info + "\t" + _formats + "\t" + _samples + "\n";
  }
};

struct gtAllele {                                        /* vcf.H:37-48, vcf.C:23-87 */
  vcfRecord *_record;
  uint32_t _pos, _refLen;
  double _qual;
  vector<char const *> _alleles;
  explicit gtAllele(vcfRecord *r) {
    _record = r;
    _pos = _record->_pos - 1;
    _refLen = (uint32_t)strlen(_record->_ref);
    _qual = _record->_qual;
    const char *s0 = _record->_arr_samples[0];
    if (s0 == nullptr) s0 = "";
    if ((strncmp(s0, "./.", 3) == 0) || (strncmp(s0, "0/0", 3) == 0)) {      /* :34-39 */
      _record->_isValid = false;
      return;
    }
    Words GT;
    GT.split(s0, "|/");                                                       /* :44 */
    _alleles.push_back(_record->_ref);                                        /* :46 */
    for (uint32_t ii = 0; ii < GT.numWords(); ii++) {                         /* :50-86 */
      int32_t altIdx = (int32_t)strtol(GT[ii], nullptr, 10);
      if (altIdx <= 0) {
        _record->_isValid = false;
        continue;
      }
      char const *hap = _record->_arr_alts[(size_t)(altIdx - 1)];
      if (hap != nullptr)
        for (uint32_t jj = 0; jj < _alleles.size(); jj++)
          if (_alleles[jj] == hap)                      /* pointer compare: same ALT listed twice */
            hap = nullptr;
      if (hap != nullptr)
        if (strcmp(_alleles[0], hap) == 0)
          hap = nullptr;
      if (hap != nullptr)
        _alleles.push_back(hap);
    }
  }
};

struct posGT {                                           /* vcf.H:57-84 */
  char const *_chr;
  uint32_t _rStart, _rEnd;
  vector<gtAllele *> _gts;
  explicit posGT(vcfRecord *record) {
    gtAllele *gt = new gtAllele(record);
    _chr = record->_chr;
    _gts.push_back(gt);
    _rStart = gt->_pos;
    _rEnd = gt->_pos + gt->_refLen;
  }
  void addGtAllele(gtAllele *gt) {
    _gts.push_back(gt);
    _rStart = min(_rStart, gt->_pos);
    _rEnd = max(_rEnd, gt->_pos + gt->_refLen);
  }
};

struct vcfFile {                                         /* vcf.H:89-125 */
  int32_t _numChr = 0;
  vector<string> _headers;
  vector<vcfRecord *> _records;
  map<string, vector<posGT *> *> _mapChrPosGT;
  uint64_t excluded = 0;

  bool loadFile(const char *inName) {                    /* vcf.C:93-149 */
    FILE *F = fopen(inName, "r");
    if (!F) return false;
    char *L = nullptr;
    size_t cap = 0;
    ssize_t n;
    while ((n = getline(&L, &cap, F)) >= 0) {
      while (n > 0 && (L[n - 1] == '\n' || L[n - 1] == '\r')) L[--n] = 0;
      if (L[0] == '#') {
        _headers.push_back(L);
        if (strncmp(L, "##contig=<ID", 12) == 0) _numChr++;
        continue;
      }
      vcfRecord *record = new vcfRecord;
      if (record->load(L) == false) {
        excluded++;
        delete record;
      } else {
        _records.push_back(record);
        string chr = record->_chr;
        if (_mapChrPosGT.count(chr) == 0) _mapChrPosGT[chr] = new vector<posGT *>;
        _mapChrPosGT[chr]->push_back(new posGT(record));
      }
    }
    free(L);
    fclose(F);
    return true;
  }

  bool mergeChrPosGT(uint32_t ksize, uint32_t comb, bool nosplit) {     /* vcf.C:156-246 */
    uint32_t K_OFFSET = 2 * ksize;
    for (auto it = _mapChrPosGT.begin(); it != _mapChrPosGT.end(); it++) {
      string chr = it->first;
      vector<posGT *> &inlist = *it->second;
      vector<posGT *> *otlist = new vector<posGT *>;
      auto byBeginCoord = [](posGT *const &A, posGT *const &B) { return (A->_rStart < B->_rStart); };
      sort(inlist.begin(), inlist.end(), byBeginCoord);
      otlist->push_back(inlist[0]);
      for (uint32_t ii = 1; ii < inlist.size(); ii++) {
        if (inlist[ii]->_gts.size() == 0) continue;
        assert(otlist->back()->_rStart <= inlist[ii]->_rStart);
        bool overlapping = (inlist[ii]->_rStart < otlist->back()->_rEnd + K_OFFSET);
        bool toomany = (otlist->back()->_gts.size() >= comb);
        if (overlapping == false) { otlist->push_back(inlist[ii]); continue; }
        if ((toomany == true) && (nosplit == false)) { otlist->push_back(inlist[ii]); continue; }
        otlist->back()->addGtAllele(inlist[ii]->_gts[0]);
      }
      delete _mapChrPosGT[chr];
      _mapChrPosGT[chr] = otlist;
    }
    return true;
  }
};

/* The lookups of varMer::score, injectable: cb != NULL replaces kmerIterator + getK(fmer, rmer) by a caller-supplied
 * function of the k-mer's TEXT (k bases, all ACGT) -- the k-agnostic form used to pin the 32 <= k <= 64 path, whose k-mers
 * do not fit the 64-bit orc_kiter / orc_lookup (tests: a plain-Python getK over arbitrary-precision k-mers,
 * oracle/plain.py).  The validity rule is kmerIterator's: a k-mer ends at a base iff the k bases up to it are ACGT. */
typedef void (*orc_getk_text_fn)(void *ctx, const char *kmer, int k, double *readK, double *asmK, double *prob);

struct Globals {
  const orc_params *p;
  const orc_lookup *R, *A;
  int reportType;
  uint32_t comb;
  orc_getk_text_fn cb = nullptr;
  void *cb_ctx = nullptr;
};

struct varMer {                                          /* varMer.H, varMer.C */
  vector<vector<int>> gtPaths;
  vector<vector<uint32_t>> idxPaths, lenPaths;
  vector<string> seqs;
  vector<uint32_t> numMs;
  vector<vector<double>> kstrs, dkstrs;
  multimap<double, int, greater<int>> avgKs;             /* sic: keys compared as ints, descending */
  posGT *posGt;
  explicit varMer(posGT *g) : posGt(g) {}

  void addSeqPath(string seq, vector<int> idxPath, vector<uint32_t> varIdxPath, vector<uint32_t> varLenPath) {   /* :37-45 */
    if (find(seqs.begin(), seqs.end(), seq) == seqs.end()) {
      seqs.push_back(seq);
      gtPaths.push_back(idxPath);
      idxPaths.push_back(varIdxPath);
      lenPaths.push_back(varLenPath);
    }
  }

  void score(Globals *g) {                               /* :48-145 */
    uint32_t numM;
    string seq;
    /* The reference leaves `prob` uninitialised; it is first WRITTEN by the first
     * valid k-mer and only multiplies |0-0| before that.  Any finite start value
     * gives the same results; we fix 1.0. */
    double prob = 1.0, readK, asmK, oDeltak, nDeltak, kMetric;
    vector<double> m_ks, m_dks;
    uint32_t idx = 0;
    const uint32_t K = (uint32_t)g->p->k;
    for (int ii = 0; ii < (int)seqs.size(); ii++) {
      numM = 0;
      seq = seqs.at(ii);
      m_ks.clear();
      m_dks.clear();
      idx = 0;
      orc_kiter kiter;
      orc_kiter_init(&kiter, g->cb ? 1 : g->p->k, seq.c_str(), seq.size());   /* cb: only its per-byte stepping is used */
      uint64_t run = 0, at = 0;                                                /* cb: valid bases ending at byte `at` */
      while (orc_kiter_next_base(&kiter)) {
        readK = 0;
        asmK = 0;
        if (g->cb) {
          run = orc_base_code((unsigned char)seq[at]) >= 0 ? run + 1 : 0;
          if (run >= K) g->cb(g->cb_ctx, seq.c_str() + at + 1 - K, (int)K, &readK, &asmK, &prob);
          at++;
        } else if (orc_kiter_is_valid(&kiter))
          orc_getK_kmers(g->p, g->R, g->A, kiter.fmer, kiter.rmer, &readK, &asmK, &prob);
        if (readK == 0)
          numM++;
        if (g->reportType == OP_FILTER) { idx++; continue; }                  /* :93-96 */
        oDeltak = std::abs(readK - asmK) * prob;                              /* :99 */
        for (int jj = 0; jj < (int)idxPaths.at(ii).size(); jj++) {            /* :103-112 */
          uint32_t idxPath = idxPaths.at(ii).at(jj);
          uint32_t lenPath = lenPaths.at(ii).at(jj);
          int gtPath = gtPaths.at(ii).at(jj);
          if (gtPath > 0 && idxPath + 1 - K <= idx && idx < idxPath + lenPath + K) {   /* uint32 arithmetic */
            asmK++;
            break;
          }
        }
        if (readK == 0) kMetric = -1;                                          /* :116-124 */
        else if (readK > asmK) kMetric = readK / asmK - 1;
        else kMetric = asmK / readK - 1;
        nDeltak = std::abs(readK - asmK) * prob;                              /* :126 */
        m_ks.push_back(kMetric);
        m_dks.push_back(oDeltak - nDeltak);
        idx++;
      }
      numMs.push_back(numM);
      kstrs.push_back(m_ks);
      dkstrs.push_back(m_dks);
    }
  }

  string getHomRecord(int idx) {                         /* :531-550 */
    string records;
    for (int i = 0; i < (int)gtPaths.at(idx).size(); i++) {
      int altIdx = gtPaths.at(idx).at(i);
      if (altIdx > 0) {
        string qualStr = to_string((int)posGt->_gts[i]->_qual);
        records = records + posGt->_chr + "\t" + to_string(posGt->_gts[i]->_pos + 1) + "\t.\t" + posGt->_gts[i]->_alleles[0] + "\t" +
                  posGt->_gts[i]->_alleles[altIdx] + "\t" + qualStr + "\t" + "PASS\t.\tGT\t1/1\n";
      }
    }
    return records;
  }

  string getHetRecord(int idx1, int idx2) {              /* :472-529 */
    string records;
    for (int i = 0; i < (int)gtPaths.at(idx1).size(); i++) {
      int altIdx1 = gtPaths.at(idx1).at(i);
      int altIdx2 = gtPaths.at(idx2).at(i);
      if (altIdx1 + altIdx2 > 0) {
        string qualStr = to_string((int)posGt->_gts[i]->_qual);
        records = records + posGt->_chr + "\t" + to_string(posGt->_gts[i]->_pos + 1) + "\t" + "." + "\t" + posGt->_gts[i]->_alleles[0] + "\t";
        if (altIdx1 == altIdx2)
          records = records + posGt->_gts[i]->_alleles[altIdx1] + "\t" + qualStr + "\t" + "PASS\t.\tGT\t1/1\n";
        else if (altIdx1 == 0 && altIdx2 > 0)
          records = records + posGt->_gts[i]->_alleles[altIdx2] + "\t" + qualStr + "\t" + "PASS\t.\tGT\t0/1\n";
        else if (altIdx1 > 0 && altIdx2 > 0)
          records = records + posGt->_gts[i]->_alleles[altIdx1] + "," + posGt->_gts[i]->_alleles[altIdx2] + "\t" + qualStr + "\t" + "PASS\t.\tGT\t1/2\n";
        else if (altIdx1 > 0 && altIdx2 == 0)
          records = records + posGt->_gts[i]->_alleles[altIdx1] + "\t" + qualStr + "\t" + "PASS\t.\tGT\t1/0\n";
      }
    }
    return records;
  }

  double getTotdK(int idx) {                             /* :647-659 */
    double sum = 0;
    vector<double> dkstr = dkstrs.at(idx);
    for (int i = 0; i < (int)dkstr.size(); i++) sum += dkstr.at(i);
    return sum;
  }
  double getMinAbsK(int idx) {                           /* :553-569 */
    double minAbsK = DBL_MAX, absK;
    vector<double> kstr = kstrs.at(idx);
    for (int i = 0; i < (int)kstr.size(); i++) {
      absK = kstr.at(i);
      if (absK < 0) continue;
      if (absK < minAbsK) minAbsK = absK;
    }
    if (minAbsK == DBL_MAX) return -1;
    return minAbsK;
  }
  double getMaxAbsK(int idx) {                           /* :572-585 */
    double maxAbsK = -2, absK;
    vector<double> kstr = kstrs.at(idx);
    for (int i = 0; i < (int)kstr.size(); i++) { absK = kstr.at(i); if (absK > maxAbsK) maxAbsK = absK; }
    return maxAbsK;
  }
  double getAvgAbsK(int idx) {                           /* :587-606 */
    double sum = 0, absK;
    vector<double> kstr = kstrs.at(idx);
    for (int i = 0; i < (int)kstr.size(); i++) { absK = kstr.at(i); if (absK >= 0) sum += absK; }
    if (kstr.size() == numMs.at(idx)) return -1;
    return sum / (kstr.size() - numMs.at(idx));
  }
  double getMedAbsK(int idx) {                           /* :608-624 */
    vector<double> kstr = kstrs.at(idx);
    sort(kstr.begin(), kstr.end());
    int i = 0;
    for (; i < (int)kstr.size(); i++) if (kstr.at(i) >= 0) break;
    if (i == (int)kstr.size()) return -1;
    return kstr.at(i + ((kstr.size() - i) / 2));
  }

  vector<vcfRecord *> bestFilter(uint32_t K) {           /* :150-199 */
    uint32_t numMissing = UINT32_MAX;
    vector<int> idxs;
    vector<vcfRecord *> records;
    for (int ii = 0; ii < (int)numMs.size(); ii++) {
      if (numMs.at(ii) == seqs.at(ii).size() - K + 1) continue;
      if (numMs.at(ii) == 0) { idxs.push_back(ii); numMissing = 0; }
      if (numMs.at(ii) < numMissing) { numMissing = numMs.at(ii); idxs.clear(); idxs.push_back(ii); }
      else if (numMs.at(ii) == numMissing) idxs.push_back(ii);
    }
    if (idxs.size() == 0) return records;
    list<int> gtIdxs;
    for (int ii = 0; ii < (int)idxs.size(); ii++) {
      int idx = idxs.at(ii);
      for (int i = 0; i < (int)gtPaths.at(idx).size(); i++)
        if (gtPaths.at(idx).at(i) > 0) gtIdxs.push_back(i);
    }
    gtIdxs.sort();
    gtIdxs.unique();
    for (list<int>::iterator it = gtIdxs.begin(); it != gtIdxs.end(); ++it)
      records.push_back(posGt->_gts[*it]->_record);
    return records;
  }

  /* betterVariant (:204-258) and strictPolish (:260-315) are the same code */
  string betterOrStrict() {
    uint32_t numMissing = UINT32_MAX;
    vector<int> idxs;
    if (numMs.size() == 0) return "";
    uint32_t refMissing = numMs.at(0);
    numMissing = refMissing;
    for (int ii = 0; ii < (int)numMs.size(); ii++) {
      if (numMs.at(ii) < numMissing) { numMissing = numMs.at(ii); idxs.clear(); idxs.push_back(ii); }
      else if (numMs.at(ii) == numMissing && numMs.at(ii) < refMissing) idxs.push_back(ii);
    }
    if (idxs.size() == 0) return "";
    int idx = idxs.at(0);
    if (idxs.size() == 1) return getHomRecord(idx);
    uint32_t seqLenMax = (uint32_t)seqs.at(idx).size();
    for (int ii = 1; ii < (int)idxs.size(); ii++) {
      uint32_t seqLen = (uint32_t)seqs.at(idxs.at(ii)).length();
      if (seqLen > seqLenMax) { seqLenMax = seqLen; idx = idxs.at(ii); }
    }
    return getHomRecord(idx);
  }

  string loosePolish(FILE *warn) {                       /* :317-395 */
    uint32_t numMissing = UINT32_MAX;
    vector<int> idxs;
    if (numMs.size() == 0) return "";
    uint32_t refMissing = numMs.at(0);
    numMissing = refMissing;
    for (int ii = 0; ii < (int)numMs.size(); ii++) {
      if (numMs.at(ii) < numMissing) { numMissing = numMs.at(ii); idxs.clear(); idxs.push_back(ii); }
      else if (numMs.at(ii) == numMissing && numMs.at(ii) <= refMissing) idxs.push_back(ii);
    }
    if (idxs.size() == 0) return "";
    int idx = idxs.at(0);
    if (idxs.size() == 1) return getHomRecord(idx);
    if (idxs.at(0) == 0 && idxs.size() == 2) return getHomRecord(idxs.at(1));
    int maxVars = 0;
    int maxIdx = idx;
    for (int ii = 1; ii < (int)idxs.size(); ii++) {
      int count = 0;
      idx = idxs.at(ii);
      for (uint64_t i = 0; i < gtPaths.at(idx).size(); i++) if (gtPaths[idx][i] > 0) count++;
      if (count > maxVars) { maxVars = count; maxIdx = idx; }
    }
    if (warn) {
      fprintf(warn, "[ WARNING ] :: Multiple (%lu) alternate pathes detected in a path beginning with variant : %s", idxs.size(), posGt->_gts[0]->_record->save().c_str());
      fprintf(warn, "[ WARNING ] :: Max. %d ALT variants selected\n", maxVars);
    }
    return getHomRecord(maxIdx);
  }

  string bestVariant(uint32_t K) {                       /* :400-467 */
    uint32_t numMissing = UINT32_MAX;
    vector<int> idxs;
    for (int ii = 0; ii < (int)numMs.size(); ii++) {
      if (numMs.at(ii) == seqs.at(ii).size() - K + 1) continue;
      if (numMs.at(ii) < numMissing) { numMissing = numMs.at(ii); idxs.clear(); idxs.push_back(ii); }
      else if (numMs.at(ii) == numMissing) idxs.push_back(ii);
    }
    if (numMissing == UINT32_MAX) return "";
    if (idxs.size() == 1) return getHomRecord(idxs.at(0));
    else if (idxs.size() > 1) {
      for (int i = 0; i < (int)idxs.size(); i++) {
        int idx = idxs.at(i);
        avgKs.insert(make_pair(getTotdK(idx), idx));
      }
      auto it = avgKs.begin();
      double avgK1 = (*it).first;
      int idx1 = (*it).second;
      it++;
      double avgK2 = (*it).first;
      int idx2 = (*it).second;
      if (avgK1 == avgK2) {
        if (seqs.at(idx1).length() >= seqs.at(idx2).length()) return getHetRecord(idx1, idx2);
        else return getHetRecord(idx2, idx1);
      } else return getHomRecord(idx1);
    }
    return "";
  }
};

/* merfin-variants.C:22-126.  Parameter passing (by value / by reference) is
 * part of the algorithm and kept exactly. */
string traverse(uint32_t idx, vector<uint32_t> &refIdxList, vector<uint32_t> refLenList,
                map<int, vector<char const *>> posHaps, string candidate, vector<int> &path, varMer *seqMer) {
  assert(idx < refIdxList.size());
  vector<char const *> &haps = posHaps[idx];
  uint32_t refLen = refLenList[idx];
  for (int j = 0; j < (int)haps.size(); j++) {
    path.push_back(j);
    char const *hap = haps[j];
    string replaced = candidate;
    int skipped = 0;
    bool overlaps = false;
    int delta = 0;
    if (j > 0) {
      refLenList[idx] = refLen;
      replaced = candidate;
      replaced.replace(refIdxList[idx], refLenList[idx], hap);
      delta = (int)strlen(hap) - (int)refLenList[idx];
      uint32_t refAffected = refIdxList[idx] + refLenList[idx];
      refLenList[idx] = (uint32_t)strlen(hap);
      for (uint32_t i = idx + 1; i < refIdxList.size(); i++) {
        if (refIdxList[i] >= refAffected) break;
        overlaps = true;
        idx++;
        path.push_back(0);
        skipped++;
      }
      if (overlaps && idx == refIdxList.size() - 1) {
        seqMer->addSeqPath(replaced, path, refIdxList, refLenList);
        for (int k = 0; k < skipped; k++) { path.pop_back(); idx--; }
        path.pop_back();
        continue;
      }
      for (uint32_t i = idx + 1; i < refIdxList.size(); i++) refIdxList[i] += delta;
    }
    if (idx + 1 < refIdxList.size())
      replaced = traverse(idx + 1, refIdxList, refLenList, posHaps, replaced, path, seqMer);
    if (idx == refIdxList.size() - 1)
      seqMer->addSeqPath(replaced, path, refIdxList, refLenList);
    for (uint32_t i = idx + 1; i < refIdxList.size(); i++) refIdxList[i] -= delta;
    for (int k = 0; k < skipped; k++) { path.pop_back(); idx--; }
    path.pop_back();
  }
  return candidate;
}

}  // namespace

/* processVariants + outputVariants over all contigs in input order.
 * mode: 4 filter, 5 polish, 6 better, 7 strict, 8 loose.  Writes the VCF text
 * (headers + selected records) to out_path; optional debug_path receives the
 * -debug lines (merfin-variants.C:240-276) as plain text; `log` receives the
 * PANIC / WARNING lines.  Returns the number of clusters evaluated, <0 on error. */
static long variants_run_impl(const orc_params *p, const orc_lookup *R, const orc_lookup *A, orc_getk_text_fn cb, void *cb_ctx, int mode, uint32_t comb,
                              int nosplit, const char *vcf_path, const char *const *names, const char *const *contigs,
                              const uint64_t *lens, uint32_t ncontigs, const char *out_path, const char *debug_path,
                              const char *log_path);

extern "C" long orc_variants_run(const orc_params *p, const orc_lookup *R, const orc_lookup *A, int mode, uint32_t comb,
                                 int nosplit, const char *vcf_path, const char *const *names, const char *const *contigs,
                                 const uint64_t *lens, uint32_t ncontigs, const char *out_path, const char *debug_path,
                                 const char *log_path) {
  return variants_run_impl(p, R, A, nullptr, nullptr, mode, comb, nosplit, vcf_path, names, contigs, lens, ncontigs, out_path, debug_path, log_path);
}

/* the same with the lookups supplied by the caller (p->k may exceed 31; p->peak and the -prob table are the callback's business) */
extern "C" long orc_variants_run_cb(const orc_params *p, orc_getk_text_fn cb, void *cb_ctx, int mode, uint32_t comb,
                                    int nosplit, const char *vcf_path, const char *const *names, const char *const *contigs,
                                    const uint64_t *lens, uint32_t ncontigs, const char *out_path, const char *debug_path,
                                    const char *log_path) {
  if (!cb) return -3;
  return variants_run_impl(p, nullptr, nullptr, cb, cb_ctx, mode, comb, nosplit, vcf_path, names, contigs, lens, ncontigs, out_path, debug_path, log_path);
}

static long variants_run_impl(const orc_params *p, const orc_lookup *R, const orc_lookup *A, orc_getk_text_fn cb, void *cb_ctx, int mode, uint32_t comb,
                              int nosplit, const char *vcf_path, const char *const *names, const char *const *contigs,
                              const uint64_t *lens, uint32_t ncontigs, const char *out_path, const char *debug_path,
                              const char *log_path) {
  vcfFile vcf;
  if (!vcf.loadFile(vcf_path)) return -1;
  vcf.mergeChrPosGT((uint32_t)p->k, comb, nosplit != 0);                     /* merfin-globals.C:216-217 */
  FILE *out = fopen(out_path, "w");
  if (!out) return -2;
  FILE *dbg = debug_path ? fopen(debug_path, "w") : nullptr;
  FILE *log = log_path ? fopen(log_path, "w") : nullptr;
  for (auto &h : vcf._headers) fprintf(out, "%s\n", h.c_str());              /* merfin-variants.C:332-333 */
  Globals G{p, R, A, mode, comb, cb, cb_ctx};
  const uint32_t K = (uint32_t)p->k;
  long nclusters = 0;
  uint64_t varMerId = 0;
  for (uint32_t c = 0; c < ncontigs; c++) {
    string result;
    auto found = vcf._mapChrPosGT.find(string(names[c]));
    if (found != vcf._mapChrPosGT.end()) {
      vector<posGT *> *posGTlist = found->second;
      vector<uint32_t> refIdxList, refLenList;
      vector<int> path;
      map<int, vector<char const *>> mapPosHap;
      const uint64_t seqLen = lens[c];
      for (uint64_t posGtIdx = 0; posGtIdx < posGTlist->size(); posGtIdx++) {
        posGT *posGt = posGTlist->at(posGtIdx);
        uint32_t rStart = posGt->_rStart, rEnd = posGt->_rEnd;
        vector<gtAllele *> &gts = posGt->_gts;
        uint32_t K_PADD = K - 1;
        if (rStart > K_PADD) rStart -= K_PADD; else rStart = 0;               /* :172-173 */
        if (rEnd < seqLen - K_PADD) rEnd += K_PADD; else rEnd = (uint32_t)seqLen;   /* :175-176 (uint64 arithmetic) */
        refIdxList.clear(); refLenList.clear(); path.clear(); mapPosHap.clear();
        for (uint32_t i = 0; i < gts.size(); i++) {
          refIdxList.push_back(gts[i]->_pos - rStart);
          refLenList.push_back(gts[i]->_refLen);
          mapPosHap[(int)i] = gts[i]->_alleles;
        }
        if (!(rStart <= rEnd && (uint64_t)rEnd <= seqLen)) {                  /* dnaSeq::copy fails, :208-211 */
          if (log) fprintf(log, "PANIC : Invalid region specified: %s : %u - %u\n", names[c], rStart, rEnd);
          continue;
        }
        string refTemplate(contigs[c] + rStart, contigs[c] + rEnd);
        if (refIdxList.size() > comb) {                                       /* :213-217 */
          if (log) fprintf(log, "PANIC : Combination %s:%u-%u has too many variants ( found %lu > %u ) to evaluate. Consider filtering the vcf upfront. Skipping...\n",
                           names[c], rStart, rEnd, gts.size(), comb);
          continue;
        }
        varMer *seqMer = new varMer(posGt);
        traverse(0, refIdxList, refLenList, mapPosHap, refTemplate, path, seqMer);
        seqMer->score(&G);
        nclusters++;
        if (dbg) {                                                            /* :240-276 */
          for (uint64_t idx = 0; idx < seqMer->seqs.size(); idx++) {
            fprintf(dbg, "%lu\t%s:%u-%u\t%s\t%u\t%.5f\t%.5f\t%.5f\t%.5f\t%.5f\t", varMerId++, names[c], rStart, rEnd,
                    seqMer->seqs[idx].c_str(), seqMer->numMs[idx],
                    seqMer->getMinAbsK((int)idx), seqMer->getMaxAbsK((int)idx), seqMer->getMedAbsK((int)idx),
                    seqMer->getAvgAbsK((int)idx), seqMer->getTotdK((int)idx));
            for (uint64_t i = 0; i < seqMer->gtPaths[idx].size(); i++) {
              int altIdx = seqMer->gtPaths[idx][i];
              if (altIdx > 0)
                fprintf(dbg, "%s %u . %s %s . PASS . GT 1/1  ", names[c], gts[i]->_pos + 1, gts[i]->_alleles[0], gts[i]->_alleles[altIdx]);
            }
            fprintf(dbg, "\n");
          }
        }
        if (mode == OP_POLISH) result += seqMer->bestVariant(K);              /* :281-304 */
        else if (mode == OP_BETTER) result += seqMer->betterOrStrict();
        else if (mode == OP_STRICT) result += seqMer->betterOrStrict();
        else if (mode == OP_LOOSE) result += seqMer->loosePolish(log);
        else {
          vector<vcfRecord *> records = seqMer->bestFilter(K);
          for (uint64_t i = 0; i < records.size(); i++) result += records[i]->save();
        }
        delete seqMer;
      }
    }
    fputs(result.c_str(), out);                                               /* outputVariants :339 */
  }
  fclose(out);
  if (dbg) fclose(dbg);
  if (log) fclose(log);
  return nclusters;
}
