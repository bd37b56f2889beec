// Force-included (-include) when the host half of the C ABI is compiled with g++ for the sanitizer run: hip_ext.h's
// hipExtLaunchKernelGGL template names two helpers that the HIP headers define for the clang HIP front end only.  They are
// declared here so that the header parses; nothing in the sanitized build calls them.
#pragma once
#include <tuple>
template <int n, typename... Ts>
void pArgs(const std::tuple<Ts...>&, void**) {}
template <typename F, typename T>
T validateArgsCountType(F, T t) { return t; }
