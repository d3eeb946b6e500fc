/* falcon9_fsw.h — C restatement of the Falcon 9 example's flight software (ascent phases).  TEST INFRASTRUCTURE ONLY:
 * see falcon9_fsw.c.  Packet layouts are the reference's (examples/falcon9/main.py:280-303 state, controller/src/main.rs:187-196 command). */
#pragma once
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { F9FSW_PAD_PRESS = 0, F9FSW_VERTICAL_RISE = 1, F9FSW_PITCH_KICK = 2, F9FSW_GRAVITY_TURN = 3, F9FSW_MECO = 4, F9FSW_FLIP = 5 };
#define F9FSW_BEYOND_ASCENT 1   /* f9fsw_step: the state machine left the phases restated here (Meco -> Flip) */

typedef struct f9fsw f9fsw;

f9fsw* f9fsw_new(void);
void f9fsw_free(f9fsw*);
/* the recorded profile the FSW flies (ELODIN_F9_PROFILE, data/<mission>/stage1_raw.json: time s, velocity m/s, altitude km) */
int f9fsw_load_profile(f9fsw*, const double* time, const double* velocity, const double* altitude_km, size_t n_raw);
int f9fsw_load_table(f9fsw*, const double* time, const double* speed, const double* alt_m, const double* vspeed, size_t n);   /* the resampled table as it is */
size_t f9fsw_profile_table(const f9fsw*, double* time, double* speed, double* alt_m, double* vspeed, size_t cap);
/* one exchange: 49-double sensor packet in, 27-double command packet out; 0, F9FSW_BEYOND_ASCENT or -1 */
int f9fsw_step(f9fsw*, const double* state49, double* cmd27);
void f9fsw_peek(const f9fsw*, double* out23);

void f9fsw_ecef_to_geodetic(const double* r, double* lat_lon_alt);
void f9fsw_quat_between(const double* from, const double* to, double* q);
double f9fsw_density(double alt_m);

#ifdef __cplusplus
}
#endif
