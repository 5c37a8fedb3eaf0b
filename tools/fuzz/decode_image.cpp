#include "ImageIO.h"
#include <cstdio>
int main(int argc, char** argv)
{
    int w, h, req;
    auto px = srx::LoadImageFromFile(argv[1], w, h, req);
    unsigned long sum = 0;
    for (auto v : px) sum = sum * 31 + v;
    printf("%d %d %d %lu %s\n", w, h, req, sum, srx::LastImageError().c_str());
}
