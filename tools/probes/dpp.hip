#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out)
{
    int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x138, 0xf, 0xf, false);        // wave_shr:1
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x130, 0xf, 0xf, false);   // wave_shl:1
    out[128 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, true);   // row_shr:1 bound_ctrl
}
int main()
{
    int *d, h[192];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int t = 0; t < 3; t++) {
        printf("%s:", t == 0 ? "wave_shr1" : t == 1 ? "wave_shl1" : "row_shr1_bc");
        for (int l = 0; l < 64; l++) if (l < 3 || (l > 13 && l < 19) || l > 61) printf(" [%d]=%d", l, h[64 * t + l]);
        printf("\n");
    }
    return 0;
}
