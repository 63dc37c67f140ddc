// signature.hpp -- Signature container (reference src/lib.rs:546-675), host only.
#pragma once
#include <string>
#include <vector>

#include "minhash.hpp"

namespace smh {

struct Signature {
  std::string klass = "sourmash_signature";  // `class`, default_class src/lib.rs:571-573
  std::string email;
  std::string hash_function = "0.murmur64";  // Default src/lib.rs:648-661
  bool has_filename = false;
  std::string filename;
  bool has_name = false;
  std::string name;
  std::string license = "CC0";               // default_license src/lib.rs:567-569
  std::vector<KmerMinHash> signatures;
  double version = 0.4;                      // default_version src/lib.rs:575-577
};

std::string sketch_md5(const KmerMinHash& mh);
bool sketch_equal(const KmerMinHash& a, const KmerMinHash& b);
bool signature_equal(const Signature& a, const Signature& b);
void signature_to_json(std::string& out, const Signature& s);
std::string signatures_to_json(const std::vector<const Signature*>& v);
std::vector<Signature> signatures_from_json(const char* data, size_t len);
std::vector<Signature> load_signatures(const char* data, size_t len, size_t ksize, const char* moltype);
std::string read_file(const std::string& path);

}  // namespace smh
