// config_dump.cpp -- prints what sxmc::load_config (sxmc_amd/include/sxmc/config.h) made of a fit configuration,
// as one JSON document: tests/test_io_cpu.py compares it field by field with sxmc_amd/io.py on the same files.
// No device call is made; built plain and with -fsanitize=address,undefined.
// Usage: config_dump <config.json> | config_dump --json <file>   (the second form only round-trips the JSON reader)
#include <cinttypes>
#include <cstdio>

#include "../../sxmc_amd/include/sxmc/config.h"

static void print_string(const std::string& s) {
  std::putchar('"');
  for (char c : s) {
    if (c == '"' || c == '\\') std::putchar('\\');
    if (c == '\n') {
      std::fputs("\\n", stdout);
      continue;
    }
    std::putchar(c);
  }
  std::putchar('"');
}

static void dump_value(const sxmc::json::Value& v) {
  using sxmc::json::Value;
  switch (v.kind) {
    case Value::Null: std::fputs("null", stdout); break;
    case Value::Bool: std::fputs(v.b ? "true" : "false", stdout); break;
    case Value::Number: std::printf("%.17g", v.num); break;
    case Value::String: print_string(v.str); break;
    case Value::Array:
      std::putchar('[');
      for (size_t i = 0; i < v.items.size(); i++) {
        if (i) std::putchar(',');
        dump_value(v.items[i]);
      }
      std::putchar(']');
      break;
    case Value::Object: {
      std::putchar('{');
      bool first = true;
      for (const auto& kv : v.members) {
        if (!first) std::putchar(',');
        first = false;
        print_string(kv.first);
        std::putchar(':');
        dump_value(kv.second);
      }
      std::putchar('}');
      break;
    }
  }
}

// order-independent fingerprints of a float table: the sum in double, in index order, and the xor of the bit patterns
static void table_summary(const std::vector<float>& t, size_t nf) {
  double sum = 0;
  uint32_t x = 0;
  for (float f : t) {
    sum += (double)f;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    x ^= u;
  }
  std::printf("{\"rows\": %zu, \"sum\": %.17g, \"xor\": %" PRIu32 "}", nf ? t.size() / nf : 0, sum, x);
}

int main(int argc, char** argv) {
  try {
    if (argc == 3 && std::string(argv[1]) == "--json") {
      dump_value(sxmc::json::parse(sxmc::detail::read_file(argv[2])));
      std::putchar('\n');
      return 0;
    }
    if (argc == 4 && std::string(argv[1]) == "--rewrite") {
      // read_table + write_chain_npz: the table of argv[2] written again, column by column, to argv[3]
      std::vector<float> m;
      std::vector<std::string> fields;
      sxmc::read_table(argv[2], m, fields);
      sxmc::write_chain_npz(argv[3], fields, m);
      return 0;
    }
    if (argc != 2) {
      std::fprintf(stderr, "usage: config_dump <config.json> | --json <file> | --rewrite <in.npz> <out.npz>\n");
      return 2;
    }
    const sxmc::FitConfig fc = sxmc::load_config(argv[1]);
    std::printf("{\"nexperiments\": %u, \"nsteps\": %u, \"error_type\": \"%s\", \"burnin_fraction\": %.9g, "
                "\"debug_mode\": %s, \"output_prefix\": \"%s\", \"seed\": %lld, \"confidence\": %.9g, \"signal_name\": \"%s\",\n",
                fc.nexperiments, fc.nsteps, fc.error_type.c_str(), (double)fc.burnin_fraction,
                fc.debug_mode ? "true" : "false", fc.output_prefix.c_str(), fc.seed, (double)fc.confidence,
                fc.signal_name.c_str());
    std::printf(" \"sample_fields\": [");
    for (size_t i = 0; i < fc.sample_fields.size(); i++) std::printf("%s\"%s\"", i ? ", " : "", fc.sample_fields[i].c_str());
    std::printf("],\n \"observables\": [");
    for (size_t i = 0; i < fc.observables.size(); i++) {
      const sxmc::Observable& o = fc.observables[i];
      std::printf("%s{\"name\": \"%s\", \"field\": \"%s\", \"field_index\": %zu, \"bins\": %zu, \"lower\": %.9g, \"upper\": %.9g}",
                  i ? ", " : "", o.name.c_str(), o.field.c_str(), o.field_index, o.bins, (double)o.lower, (double)o.upper);
    }
    std::printf("],\n \"cuts\": [");
    for (size_t i = 0; i < fc.cuts.size(); i++) {
      const sxmc::Observable& o = fc.cuts[i];
      std::printf("%s{\"name\": \"%s\", \"field\": \"%s\", \"lower\": %.9g, \"upper\": %.9g}", i ? ", " : "",
                  o.name.c_str(), o.field.c_str(), (double)o.lower, (double)o.upper);
    }
    std::printf("],\n \"systematics\": [");
    for (size_t i = 0; i < fc.systematics.size(); i++) {
      const sxmc::Systematic& s = fc.systematics[i];
      std::printf("%s{\"name\": \"%s\", \"type\": %d, \"observable_field_index\": %zu, \"truth_field_index\": %zu, "
                  "\"npars\": %zu, \"fixed\": %s, \"pidx\": [", i ? ", " : "", s.name.c_str(), (int)s.type,
                  s.observable_field_index, s.truth_field_index, s.npars, s.fixed ? "true" : "false");
      for (size_t k = 0; k < s.pidx.size(); k++) std::printf("%s%d", k ? ", " : "", (int)s.pidx[k]);
      std::printf("], \"means\": [");
      for (size_t k = 0; k < s.means.size(); k++) std::printf("%s%.17g", k ? ", " : "", s.means[k]);
      std::printf("], \"sigmas\": [");
      for (size_t k = 0; k < s.sigmas.size(); k++) std::printf("%s%.17g", k ? ", " : "", s.sigmas[k]);
      std::printf("]}");
    }
    std::printf("],\n \"sources\": [");
    for (size_t i = 0; i < fc.sources.size(); i++) {
      const sxmc::Source& s = fc.sources[i];
      std::printf("%s{\"name\": \"%s\", \"index\": %zu, \"mean\": %.9g, \"sigma\": %.9g, \"fixed\": %s}", i ? ", " : "",
                  s.name.c_str(), s.index, (double)s.mean, (double)s.sigma, s.fixed ? "true" : "false");
    }
    std::printf("],\n \"signals\": [");
    for (size_t i = 0; i < fc.signals.size(); i++) {
      const sxmc::Signal& s = fc.signals[i];
      std::printf("%s{\"name\": \"%s\", \"dataset\": %u, \"source_index\": %zu, \"nexpected\": %.17g, \"n_mc\": %zu, "
                  "\"table\": ", i ? ",\n   " : "", s.name.c_str(), s.dataset, s.source.index, s.nexpected, s.n_mc);
      table_summary(fc.tables[i], fc.nfields);
      std::printf("}");
    }
    std::printf("],\n \"data\": {");
    bool first = true;
    for (const auto& kv : fc.data) {
      std::printf("%s\"%u\": [", first ? "" : ", ", kv.first);
      first = false;
      for (size_t f = 0; f < kv.second.size(); f++) {   // one table per listed file (experiment f fits file f)
        if (f) std::printf(", ");
        table_summary(kv.second[f], fc.observables.size() + 1);
      }
      std::printf("]");
    }
    std::printf("},\n \"same_systematics_everywhere\": %s}\n", sxmc::same_systematics_everywhere(fc) ? "true" : "false");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "config_dump: %s\n", e.what());
    return 1;
  }
}
