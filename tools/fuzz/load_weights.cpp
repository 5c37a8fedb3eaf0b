#include "srx_weights.h"
#include <cstdio>
int main(int argc, char** argv)
{
    srx::ML::WeightMapType m;
    const bool ok = srx::LoadWeights(argv[1], m, false);
    srx::FoldedWeights f;
    const bool ok2 = ok && srx::FoldWeights(m, f);
    printf("%d %d %zu\n", int(ok), int(ok2), m.size());
}
