// TEST INFRASTRUCTURE: runs the CPU build of the PNG / DEFLATE core (png_core_host.cpp) over every file named on the command
// line and prints one status per file.  tests/test_cpu_png.py compiles this with -fsanitize=address,undefined and feeds it
// damaged files: every one must come back with a status (0 or an error code), never with a sanitizer report.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -o oracle/_host/png_fuzz oracle/png_fuzz_main.cpp
#include "png_core_host.cpp"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int H = atoi(argv[1]), W = atoi(argv[2]);
    std::vector<uint8_t> out((size_t)H * W * 3);
    for (int a = 3; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) return 3;
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        // exact-size heap block: a read one byte past the file is an AddressSanitizer report
        uint8_t* data = (uint8_t*)malloc(n > 0 ? n : 1);
        if (n > 0 && fread(data, 1, n, f) != (size_t)n) return 4;
        fclose(f);
        const int rc = sc_png_host_decode(data, n, out.data(), H, W);
        printf("%d\n", rc);
        free(data);
    }
    return 0;
}
