// intervals_dump.cpp -- sxmc::contour_intervals / sxmc::projection_intervals (sxmc_amd/include/sxmc/ensemble.h: the
// C++ forms of Contour::get_interval, contour.cpp:30-69, and Projection::get_interval, projection.cpp:14-77) on a
// chain read from a file, printed as JSON: tests/test_intervals.py compares them with the Python forms and with the
// brute-force restatement in oracle/intervals.py on the same chain.  No device call.
// Usage: intervals_dump <chain.f32> <ncolumns> <cl>      (row-major float32, last column = likelihood)
//        intervals_dump <chain.f32> <ncolumns> <cl> --report name0,name1,...   the text of
//            LikelihoodSpace::print_best_fit + print_correlations (likelihood.cpp:34-72) for that chain with those
//            parameter names (contour intervals), as sxmc.cpp:100-101 prints them per experiment
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../sxmc_amd/include/sxmc/ensemble.h"

int main(int argc, char** argv) {
  if (argc != 4 && !(argc == 6 && std::string(argv[4]) == "--report")) {
    std::fprintf(stderr, "usage: intervals_dump <chain.f32> <ncolumns> <cl> [--report name0,name1,...]\n");
    return 2;
  }
  try {
    const size_t ncol = (size_t)std::atoi(argv[2]);
    const float cl = (float)std::atof(argv[3]);
    std::ifstream f(argv[1], std::ios::binary);
    if (!f || ncol < 2) throw std::runtime_error("cannot read the chain");
    sxmc::Chain chain;
    f.seekg(0, std::ios::end);
    const size_t bytes = (size_t)f.tellg();
    f.seekg(0);
    chain.rows.resize(bytes / 4);
    f.read(reinterpret_cast<char*>(chain.rows.data()), (std::streamsize)(chain.rows.size() * 4));
    for (size_t i = 0; i + 1 < ncol; i++) chain.names.push_back("p" + std::to_string(i));
    chain.names.push_back("likelihood");
    if (argc == 6) {
      std::string names = argv[5];
      for (size_t i = 0, at = 0; i + 1 < ncol; i++) {
        const size_t comma = names.find(',', at);
        chain.names[i] = names.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
        at = comma == std::string::npos ? names.size() : comma + 1;
      }
      sxmc::print_best_fit(std::cout, chain, sxmc::contour_intervals(chain, cl));
      sxmc::print_correlations(std::cout, chain);
      sxmc::print_best_fit(std::cout, chain, sxmc::contour_intervals(chain, cl));   // (the precision set above sticks, as in the reference)
      return 0;
    }
    const std::vector<sxmc::Interval> c = sxmc::contour_intervals(chain, cl), p = sxmc::projection_intervals(chain, cl);
    auto num = [](float x) {   // (JSON has no inf / nan: a fit that ran away prints as null)
      char b[40];
      if (std::isfinite(x)) std::snprintf(b, sizeof b, "%.9g", (double)x);
      else std::snprintf(b, sizeof b, "null");
      return std::string(b);
    };
    auto dump = [&](const char* name, const std::vector<sxmc::Interval>& v) {
      std::printf("\"%s\": [", name);
      for (size_t i = 0; i < v.size(); i++) {
        std::printf("%s[%s, %s, %s, %s, %s]", i ? ", " : "", num(v[i].point_estimate).c_str(), num(v[i].lower).c_str(),
                    num(v[i].upper).c_str(), num(v[i].coverage).c_str(), v[i].one_sided ? "true" : "false");
      }
      std::printf("]");
    };
    std::printf("{");
    dump("contour", c);
    std::printf(", ");
    dump("projection", p);
    std::printf("}\n");
    return 0;
  } catch (const pdfz::Error& e) {
    std::fprintf(stderr, "intervals_dump: %s\n", e.msg.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "intervals_dump: %s\n", e.what());
  }
  return 1;
}
