// edgeconv_layout.h -- offsets inside the packed EdgeConv parameter block (l3d_edgeconv_pack).
#pragma once
#define EC_C1 64
#define EC_C2 64
#define EC_C3 128
#define EC_C4 256
#define EC_OFF_W1 0
#define EC_OFF_W2 (EC_OFF_W1 + 8 * EC_C1)
#define EC_OFF_W3 (EC_OFF_W2 + EC_C1 * EC_C2)
#define EC_OFF_W4 (EC_OFF_W3 + EC_C2 * EC_C3)
#define EC_OFF_B1 (EC_OFF_W4 + EC_C3 * EC_C4)
#define EC_OFF_B2 (EC_OFF_B1 + EC_C1)
#define EC_OFF_B3 (EC_OFF_B2 + EC_C2)
#define EC_OFF_B4 (EC_OFF_B3 + EC_C3)
#define EC_PACKED_V1_FLOATS (EC_OFF_B4 + EC_C4)

// second weight copy for the register-chained kernel (edgeconv2.hip), see l3d_edgeconv_pack
#define EC2_OFF_W1 EC_PACKED_V1_FLOATS
#define EC2_OFF_W2 (EC2_OFF_W1 + 8 * EC_C1)
#define EC2_OFF_W3 (EC2_OFF_W2 + EC_C1 * EC_C2)
#define EC2_OFF_W4 (EC2_OFF_W3 + EC_C2 * EC_C3)
#define EC_PACKED_FLOATS (EC2_OFF_W4 + EC_C3 * EC_C4)
