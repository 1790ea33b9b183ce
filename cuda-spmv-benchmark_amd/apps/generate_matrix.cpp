// generate_matrix -- counterpart of reference src/matrix/generate_matrix.cu:
//   generate_matrix <n> <output.mtx>     writes the n x n 5-point stencil (centre 5.0, neighbours -1.0)
// in Matrix Market coordinate format with the "% STENCIL_GRID_SIZE n" comment the operators read.
#include "app_common.hpp"

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "Usage: %s <grid_size> <output_file.mtx>\n", argv[0]);
        return 1;
    }
    const int n = atoi(argv[1]);
    if (n <= 0) {
        fprintf(stderr, "Error: grid size must be positive\n");
        return 1;
    }
    printf("Generating %dx%d 5-point stencil matrix (%lld unknowns)...\n", n, n, (long long)n * n);
    return write_matrix_market_stencil5(n, argv[2]);
}
