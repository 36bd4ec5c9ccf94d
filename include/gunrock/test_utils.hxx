// gunrock/test_utils.hxx -- harness helpers the reference's enactors and test drivers include
// (gunrock/tests/test_utils.hxx): CommandLineArgs (--key=value / --flag), display_device_data,
// test_timer_t, validate (int: exact, float: |a-b| < 0.01).
#pragma once
#include <chrono>
#include <cmath>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../mgx/runtime.hpp"

using std::cout;
using std::endl;

class CommandLineArgs {
  std::map<std::string, std::string> pairs;

 public:
  CommandLineArgs(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
      std::string arg = argv[i];
      if (arg.size() < 3 || arg[0] != '-' || arg[1] != '-') continue;
      const size_t eq = arg.find('=');
      if (eq == std::string::npos) pairs[arg.substr(2)] = "";
      else pairs[arg.substr(2, eq - 2)] = arg.substr(eq + 1);
    }
  }
  bool CheckCmdLineFlag(const char* name) const { return pairs.count(name) != 0; }
  template <typename T>
  void GetCmdLineArgument(const char* name, T& val) const {
    auto it = pairs.find(name);
    if (it == pairs.end()) return;
    std::istringstream ss(it->second);
    ss >> val;
  }
  // comma separated list
  template <typename T>
  void GetCmdLineArguments(const char* name, std::vector<T>& vals) const {
    auto it = pairs.find(name);
    if (it == pairs.end()) return;
    vals.clear();
    std::istringstream ss(it->second);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
      std::istringstream ts(tok);
      T v;
      ts >> v;
      vals.push_back(v);
    }
  }
  int ParsedArgc() const { return (int)pairs.size(); }
};
template <>
inline void CommandLineArgs::GetCmdLineArgument<std::string>(const char* name, std::string& val) const {
  auto it = pairs.find(name);
  if (it != pairs.end()) val = it->second;
}

template <typename type_t>
hipError_t display_device_data(const type_t* data, std::size_t length) {
  std::vector<type_t> dest(length);
  hipError_t ret = mgx::dtoh(dest, data, length);
  if (ret != hipSuccess) return ret;
  for (const auto& item : dest) std::cout << item << ' ';
  std::cout << std::endl;
  return ret;
}

class test_timer_t {
  std::chrono::time_point<std::chrono::steady_clock> s;
  bool counting = false;

 public:
  void start() {
    if (!counting) { counting = true; s = std::chrono::steady_clock::now(); }
  }
  double end() {
    if (!counting) return 0.0;
    std::chrono::duration<double> elapsed = std::chrono::steady_clock::now() - s;
    return elapsed.count();
  }
};

inline bool validate(std::vector<int>& gpu_vals, std::vector<int>& cpu_vals) {
  if (gpu_vals.size() != cpu_vals.size()) return false;
  for (size_t i = 0; i < gpu_vals.size(); ++i)
    if (gpu_vals[i] != cpu_vals[i]) return false;
  return true;
}
inline bool validate(std::vector<float>& gpu_vals, std::vector<float>& cpu_vals) {
  if (gpu_vals.size() != cpu_vals.size()) return false;
  for (size_t i = 0; i < gpu_vals.size(); ++i)
    if (std::fabs(gpu_vals[i] - cpu_vals[i]) >= 0.01f) return false;
  return true;
}
