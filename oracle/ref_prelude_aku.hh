// ref_prelude_aku.hh -- forced include (g++ -include) for reference sources that predate `namespace aku`.
//
// aku/tests/random_feature_test.cc names FeatureGenerator / FeatureVec / FeatureBuffer without a
// using-directive; the reference's own headers declare them inside `namespace aku`
// (aku/FeatureGenerator.hh:11, aku/FeatureBuffer.hh:11), so that file does not compile against the reference's
// headers as they stand either.  This prelude supplies the one line the tools of the same directory carry
// (aku/feacat.cc:8, aku/phone_probs.cc:21); the test's text itself is compiled unchanged from where it lies.
#pragma once
#include "FeatureGenerator.hh"
using namespace aku;
