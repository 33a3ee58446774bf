// strm_main.cpp -- `strm arg...`: each argument becomes one line on stdout
// (reference tool src/strm/Strm.cpp:18-35).
#include <cstdio>
#include <cstring>

int main(int argc, const char* argv[])
{
    for (int i = 1; i < argc; ++i) {
        fputs(argv[i], stdout);
        fputc('\n', stdout);
    }
    fflush(stdout);
    return 0;
}
