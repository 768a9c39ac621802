/* oracle/vp8_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's (excamera/alfalfa, /root/reference/src/decoder) VP8
 * decode path, written from the algorithm, single-threaded, one frame at a time, straight
 * raster-order loops.  It exists to CHECK the HIP path; the product never links or calls it.
 *
 * Parity pin: tests/test_oracle_vs_ref.py compares every output raster (all three padded
 * planes, hidden frames included) with oracle/_ref/ref_decode (the reference compiled in
 * place) on encoder-generated and synthesised streams; tests/golden/ holds reference outputs.
 *
 * Each section cites the reference file:line it restates.  Quirks Q1..Q11 are SURVEY.md 8a.
 */
#include "vp8_oracle.h"
#include "vp8_tables.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { DC_PRED, V_PRED, H_PRED, TM_PRED, B_PRED, NEARESTMV, NEARMV, ZEROMV, NEWMV, SPLITMV };
enum { B_DC_PRED, B_TM_PRED, B_VE_PRED, B_HE_PRED, B_LD_PRED, B_RD_PRED, B_VR_PRED, B_VL_PRED,
       B_HD_PRED, B_HU_PRED, LEFT4X4, ABOVE4X4, ZERO4X4, NEW4X4 };
enum { REF_CURRENT, REF_LAST, REF_GOLDEN, REF_ALT };
enum { BT_Y_AFTER_Y2 = 0, BT_Y2 = 1, BT_UV = 2, BT_Y_NO_Y2 = 3 }; /* block.hh:46 */

/* ---- trees, RFC 6386 form as in modemv_data.cc:162-281 (leaf = -value, inner = index) ---- */
static const int8_t kf_y_mode_tree[8] = { -B_PRED, 2, 4, 6, -DC_PRED, -V_PRED, -H_PRED, -TM_PRED };
static const int8_t y_mode_tree[8] = { -DC_PRED, 2, 4, 6, -V_PRED, -H_PRED, -TM_PRED, -B_PRED };
static const int8_t uv_mode_tree[6] = { -DC_PRED, 2, -V_PRED, 4, -H_PRED, -TM_PRED };
static const int8_t b_mode_tree[18] = { -B_DC_PRED, 2, -B_TM_PRED, 4, -B_VE_PRED, 6, 8, 12, -B_HE_PRED, 10,
                                        -B_RD_PRED, -B_VR_PRED, -B_LD_PRED, 14, -B_VL_PRED, 16,
                                        -B_HD_PRED, -B_HU_PRED };
static const int8_t small_mv_tree[14] = { 2, 8, 4, 6, -0, -1, -2, -3, 10, 12, -4, -5, -6, -7 };
static const int8_t mv_ref_tree[8] = { -ZEROMV, 2, -NEARESTMV, 4, -NEARMV, 6, -NEWMV, -SPLITMV };
static const int8_t submv_ref_tree[6] = { -LEFT4X4, 2, -ABOVE4X4, 4, -ZERO4X4, -NEW4X4 };
static const int8_t split_mv_tree[6] = { -3, 2, -2, 4, -0, -1 };
static const int8_t segment_id_tree[6] = { 2, 4, -0, -1, -2, -3 };

static const uint8_t zigzag[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };       /* tokens.hh:55 */
static const uint8_t coeff_band[16] = { 0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7 };          /* tokens.hh:54 */
static const int16_t sixtap[8][6] = { { 0, 0, 128, 0, 0, 0 },   { 0, -6, 123, 12, -1, 0 },          /* prediction.cc:645-653 */
                                      { 2, -11, 108, 36, -8, 1 }, { 0, -9, 93, 50, -6, 0 },
                                      { 3, -16, 77, 77, -16, 3 }, { 0, -6, 50, 93, -9, 0 },
                                      { 1, -8, 36, 108, -11, 2 }, { 0, -1, 12, 123, -6, 0 } };

/* ---------------- boolean entropy decoder: bool_decoder.hh:45-120 (dixie form) ---------------- */
typedef struct { const uint8_t * p, * end; uint32_t range, value; int bit_count; } booldec;

static void bd_load( booldec * d ) { if ( d->p < d->end ) d->value |= *d->p++; } /* past end -> zeros (:56-65) */
static void bd_init( booldec * d, const uint8_t * p, size_t n )
{
  d->p = p; d->end = p + n; d->range = 255; d->value = 0; d->bit_count = 0;
  bd_load( d ); d->value <<= 8; bd_load( d );
}
static int bd_get( booldec * d, int prob )
{
  const uint32_t split = 1 + ( ( ( d->range - 1 ) * (uint32_t) prob ) >> 8 );
  const uint32_t SPLIT = split << 8;
  int ret;
  if ( d->value >= SPLIT ) { ret = 1; d->range -= split; d->value -= SPLIT; }
  else { ret = 0; d->range = split; }
  while ( d->range < 128 ) {
    d->value <<= 1; d->range <<= 1;
    if ( ++d->bit_count == 8 ) { d->bit_count = 0; bd_load( d ); }
  }
  return ret;
}
static int bd_flag( booldec * d ) { return bd_get( d, 128 ); }
static int bd_uint( booldec * d, int bits ) /* MSB first: vp8_header_structures.hh:52-66 */
{ int v = 0; while ( bits-- ) v = ( v << 1 ) | bd_get( d, 128 ); return v; }
static int bd_sint( booldec * d, int bits ) /* magnitude then sign: :71-84 */
{ const int v = bd_uint( d, bits ); return bd_flag( d ) ? -v : v; }
static int bd_tree( booldec * d, const int8_t * tree, const uint8_t * probs ) /* tree.cc:35-57 */
{ int i = 0; while ( ( i = tree[ i + bd_get( d, probs[ i >> 1 ] ) ] ) > 0 ) {} return -i; }

/* ---------------- persistent decoder state: decoder.hh:57-225 ---------------- */
typedef struct {
  uint8_t coeff[4][8][3][11];
  uint8_t y_mode[4], uv_mode[3];
  uint8_t mv[2][19];
} probtab;

typedef struct { int enabled, absolute; int8_t quant[4], lf[4]; uint8_t * map; } segmentation;
typedef struct { int enabled; int8_t ref[4], mode[4]; } filteradj;

typedef struct { uint8_t * plane[3]; } raster;

typedef struct {
  int key, show;
  int seg_enabled, seg_update_map, seg_update_data, seg_abs;
  int seg_quant[4], seg_lf[4];
  uint8_t seg_tree_probs[3];
  int filter_type, lf_level, sharpness;
  int lf_adj_enabled, lf_delta_update, ref_delta[4], mode_delta[4];
  int log2_parts;
  int y_ac_qi, y_dc, y2_dc, y2_ac, uv_dc, uv_ac;
  int refresh_entropy, refresh_last, refresh_golden, refresh_alt, copy_golden, copy_alt;
  int sign_bias_golden, sign_bias_alt;
  int skip_enabled, prob_skip;
  int prob_inter, prob_last, prob_golden;
} frame_header;

struct vp8o_decoder {
  int width, height, mbw, mbh, pw, ph;    /* display dims, MB dims, padded luma dims */
  probtab probs;                           /* persistent */
  segmentation seg;
  filteradj fadj;
  raster ref[4];                           /* [1] last [2] golden [3] alt ; [0] = current output */
  frame_header hdr;
  vp8o_mb * mbs;
  uint8_t * above_nz;                      /* per MB column: 4 Y, 2 U, 2 V, 1 Y2 = 9 flags */
  int phases;
  int conceal;                             /* Decoder::set_error_concealment (decoder.hh:298) */
  char err[160];
};

static int fail( vp8o_decoder * d, int code, const char * msg )
{ snprintf( d->err, sizeof d->err, "%s", msg ); return code; }

static void raster_alloc( raster * r, int pw, int ph )
{
  r->plane[0] = (uint8_t *) calloc( (size_t) pw * ph, 1 );
  r->plane[1] = (uint8_t *) calloc( (size_t) ( pw / 2 ) * ( ph / 2 ), 1 );
  r->plane[2] = (uint8_t *) calloc( (size_t) ( pw / 2 ) * ( ph / 2 ), 1 );
}
static void raster_copy( vp8o_decoder * d, int dst, int src )
{
  if ( dst == src ) return;
  memcpy( d->ref[dst].plane[0], d->ref[src].plane[0], (size_t) d->pw * d->ph );
  memcpy( d->ref[dst].plane[1], d->ref[src].plane[1], (size_t) ( d->pw / 2 ) * ( d->ph / 2 ) );
  memcpy( d->ref[dst].plane[2], d->ref[src].plane[2], (size_t) ( d->pw / 2 ) * ( d->ph / 2 ) );
}

static void probs_default( probtab * p ) /* decoder.hh:59-70 defaults */
{
  memcpy( p->coeff, vp8o_default_coeff_probs, sizeof p->coeff );
  memcpy( p->y_mode, vp8o_default_y_mode_probs, 4 );
  memcpy( p->uv_mode, vp8o_default_uv_mode_probs, 3 );
  memcpy( p->mv, vp8o_default_mv_probs, sizeof p->mv );
}

vp8o_decoder * vp8o_create( int width, int height )
{
  vp8o_decoder * d = (vp8o_decoder *) calloc( 1, sizeof *d );
  d->width = width; d->height = height;
  d->mbw = ( width + 15 ) / 16; d->mbh = ( height + 15 ) / 16;  /* vp8_raster.hh:286 */
  d->pw = 16 * d->mbw; d->ph = 16 * d->mbh;                     /* prediction.cc:94-97 */
  probs_default( &d->probs );
  for ( int i = 0; i < 4; i++ ) raster_alloc( &d->ref[i], d->pw, d->ph );
  d->seg.map = (uint8_t *) malloc( (size_t) d->mbw * d->mbh );
  d->mbs = (vp8o_mb *) calloc( (size_t) d->mbw * d->mbh, sizeof( vp8o_mb ) );
  d->above_nz = (uint8_t *) calloc( (size_t) d->mbw, 9 );
  d->phases = 3;
  return d;
}
void vp8o_destroy( vp8o_decoder * d )
{
  if ( !d ) return;
  for ( int i = 0; i < 4; i++ ) for ( int p = 0; p < 3; p++ ) free( d->ref[i].plane[p] );
  free( d->seg.map ); free( d->mbs ); free( d->above_nz ); free( d );
}
void vp8o_set_error_concealment( vp8o_decoder * d, int on ) { d->conceal = on != 0; }

const char * vp8o_error( const vp8o_decoder * d ) { return d->err; }
void vp8o_set_phases( vp8o_decoder * d, int mask ) { d->phases = mask; }
const uint8_t * vp8o_plane( const vp8o_decoder * d, int plane, int * w, int * h )
{
  if ( w ) *w = plane ? d->pw / 2 : d->pw;
  if ( h ) *h = plane ? d->ph / 2 : d->ph;
  return d->ref[0].plane[plane];
}
const uint8_t * vp8o_ref_plane( const vp8o_decoder * d, int which, int plane ) { return d->ref[which].plane[plane]; }
const vp8o_mb * vp8o_macroblocks( const vp8o_decoder * d, int * mbw, int * mbh )
{ if ( mbw ) *mbw = d->mbw; if ( mbh ) *mbh = d->mbh; return d->mbs; }
void vp8o_get_probs( const vp8o_decoder * d, uint8_t out[1101] )
{
  memcpy( out, d->probs.coeff, 1056 ); memcpy( out + 1056, d->probs.y_mode, 4 );
  memcpy( out + 1060, d->probs.uv_mode, 3 ); memcpy( out + 1063, d->probs.mv, 38 );
}
void vp8o_get_frame_info( const vp8o_decoder * d, vp8o_frame_info * o )
{
  const frame_header * h = &d->hdr;
  o->key_frame = h->key; o->shown = h->show; o->loop_filter_level = h->lf_level; o->sharpness = h->sharpness;
  o->num_partitions = 1 << h->log2_parts; o->segmentation_enabled = d->seg.enabled;
  o->filter_adjustments_enabled = d->fadj.enabled; o->q_index = h->y_ac_qi;
  o->refresh_last = h->refresh_last; o->refresh_golden = h->refresh_golden; o->refresh_alt = h->refresh_alt;
  o->copy_to_golden = h->copy_golden; o->copy_to_alt = h->copy_alt;
}

/* ---------------- frame header: frame_header.hh:37-295 ---------------- */
static int read_delta_q( booldec * bd ) { return bd_flag( bd ) ? bd_sint( bd, 4 ) : 0; } /* Flagged<Signed<4>> */

static void parse_segmentation_and_filter( booldec * bd, frame_header * h )
{
  /* Flagged<UpdateSegmentation>: frame_header.hh:104-126 */
  h->seg_enabled = bd_flag( bd );
  h->seg_update_map = h->seg_update_data = 0;
  if ( h->seg_enabled ) {
    h->seg_update_map = bd_flag( bd );
    h->seg_update_data = bd_flag( bd );
    if ( h->seg_update_data ) {
      h->seg_abs = bd_flag( bd );
      for ( int i = 0; i < 4; i++ ) h->seg_quant[i] = bd_flag( bd ) ? bd_sint( bd, 7 ) : 0; /* get_or(0): decoder_state.hh:46 */
      for ( int i = 0; i < 4; i++ ) h->seg_lf[i] = bd_flag( bd ) ? bd_sint( bd, 6 ) : 0;
    }
    if ( h->seg_update_map )
      for ( int i = 0; i < 3; i++ ) h->seg_tree_probs[i] = bd_flag( bd ) ? bd_uint( bd, 8 ) : 255; /* frame.cc:88 */
  }
  h->filter_type = bd_flag( bd );
  h->lf_level = bd_uint( bd, 6 );
  h->sharpness = bd_uint( bd, 3 );
  /* Flagged<Flagged<ModeRefLFDeltaUpdate>>: frame_header.hh:65-79 */
  h->lf_adj_enabled = bd_flag( bd );
  h->lf_delta_update = 0;
  if ( h->lf_adj_enabled ) {
    h->lf_delta_update = bd_flag( bd );
    if ( h->lf_delta_update ) {
      for ( int i = 0; i < 4; i++ ) h->ref_delta[i] = bd_flag( bd ) ? bd_sint( bd, 6 ) : 0;  /* get_or(0): decoder_state.hh:60-61 */
      for ( int i = 0; i < 4; i++ ) h->mode_delta[i] = bd_flag( bd ) ? bd_sint( bd, 6 ) : 0;
    }
  }
  h->log2_parts = bd_uint( bd, 2 );
  h->y_ac_qi = bd_uint( bd, 7 );                       /* QuantIndices: frame_header.hh:37-49 */
  h->y_dc = read_delta_q( bd ); h->y2_dc = read_delta_q( bd ); h->y2_ac = read_delta_q( bd );
  h->uv_dc = read_delta_q( bd ); h->uv_ac = read_delta_q( bd );
}

static void parse_token_prob_updates( booldec * bd, probtab * p ) /* frame_header.hh:128-145, probability_tables.cc:72-89 */
{
  for ( int i = 0; i < 4; i++ ) for ( int j = 0; j < 8; j++ ) for ( int k = 0; k < 3; k++ ) for ( int l = 0; l < 11; l++ )
    if ( bd_get( bd, vp8o_coeff_update_probs[ ( ( i * 8 + j ) * 3 + k ) * 11 + l ] ) )
      p->coeff[i][j][k][l] = (uint8_t) bd_uint( bd, 8 );
}

/* ---------------- per-frame parsing helpers ---------------- */
typedef struct { int16_t x, y; } mvec;

static int mv_read_component( booldec * bd, const uint8_t * p ) /* macroblock.cc:198-229 */
{
  enum { IS_SHORT, SIGN, SHORT, BITS = SHORT + 8 - 1, LONG_WIDTH = 10 };
  int x = 0;
  if ( bd_get( bd, p[IS_SHORT] ) ) {
    for ( int i = 0; i < 3; i++ ) x += bd_get( bd, p[BITS + i] ) << i;
    for ( int i = LONG_WIDTH - 1; i > 3; i-- ) x += bd_get( bd, p[BITS + i] ) << i;
    if ( !( x & 0xFFF0 ) || bd_get( bd, p[BITS + 3] ) ) x += 8;
  } else {
    x = bd_tree( bd, small_mv_tree, p + SHORT );
  }
  x <<= 1;
  if ( x && bd_get( bd, p[SIGN] ) ) x = -x;
  return (int16_t) x;
}
static mvec mv_read( booldec * bd, const probtab * p ) /* y first: macroblock.cc:283-287 */
{ mvec m; m.y = (int16_t) mv_read_component( bd, p->mv[0] ); m.x = (int16_t) mv_read_component( bd, p->mv[1] ); return m; }

static int clampi( int v, int lo, int hi ) { return v < lo ? lo : ( v > hi ? hi : v ); }

static mvec mv_clamp( mvec m, int col, int row, int mbw, int mbh ) /* Scorer::clamp macroblock.cc:183-195 */
{
  const int to_left = clampi( -( ( col * 16 ) << 3 ) - 128, -32768, 32767 );
  const int to_right = clampi( ( ( ( mbw - 1 - col ) * 16 ) << 3 ) + 128, -32768, 32767 );
  const int to_top = clampi( -( ( row * 16 ) << 3 ) - 128, -32768, 32767 );
  const int to_bottom = clampi( ( ( ( mbh - 1 - row ) * 16 ) << 3 ) + 128, -32768, 32767 );
  m.x = (int16_t) clampi( m.x, to_left, to_right );
  m.y = (int16_t) clampi( m.y, to_top, to_bottom );
  return m;
}

typedef struct { mvec best, nearest, near; uint8_t ctx[4]; } census;

/* Scorer: scorer.hh:35-78, macroblock.cc:143-181, 301-312 */
static void census_run( const vp8o_decoder * d, int col, int row, int flipped, const uint8_t * mb_flipped, census * out )
{
  uint8_t score[4] = { 0, 0, 0, 0 };
  mvec mvs[4]; memset( mvs, 0, sizeof mvs );
  int idx = 0, split_score = 0;
  const int nb_col[3] = { col, col - 1, col - 1 }, nb_row[3] = { row - 1, row, row - 1 };
  const int weight[3] = { 2, 2, 1 };   /* above, left, above-left */
  for ( int n = 0; n < 3; n++ ) {
    if ( nb_col[n] < 0 || nb_row[n] < 0 ) continue;
    const int ni = nb_row[n] * d->mbw + nb_col[n];
    const vp8o_mb * mb = &d->mbs[ni];
    if ( mb->ref_frame == REF_CURRENT ) continue;
    mvec mv; mv.x = mb->mv[15][0]; mv.y = mb->mv[15][1];   /* base MV = Y_.at(3,3): macroblock.cc:120-129 */
    if ( mb_flipped[ni] != flipped ) { mv.x = (int16_t) -mv.x; mv.y = (int16_t) -mv.y; }
    if ( mv.x == 0 && mv.y == 0 ) score[0] += weight[n];
    else {
      if ( !( mv.x == mvs[idx].x && mv.y == mvs[idx].y ) ) { idx++; mvs[idx] = mv; }
      score[idx] += weight[n];
    }
    if ( mb->y_mode == SPLITMV ) split_score += weight[n];
  }
  /* calculate(): Q8 */
  if ( score[3] ) { if ( mvs[idx].x == mvs[1].x && mvs[idx].y == mvs[1].y ) score[1] += score[3]; }
  if ( score[2] > score[1] ) {
    uint8_t t = score[1]; score[1] = score[2]; score[2] = t;
    mvec tm = mvs[1]; mvs[1] = mvs[2]; mvs[2] = tm;
  }
  if ( score[1] >= score[0] ) mvs[0] = mvs[1];
  out->best = mvs[0]; out->nearest = mvs[1]; out->near = mvs[2];
  out->ctx[0] = score[0]; out->ctx[1] = score[1]; out->ctx[2] = score[2]; out->ctx[3] = (uint8_t) split_score;
}

static int16_t chroma_round( int v ) { return (int16_t) ( v >= 0 ? ( v + 4 ) >> 3 : -( ( -v + 4 ) >> 3 ) ); } /* macroblock.cc:289-299 */

/* partition layouts mv_partitions: modemv_data.cc:245-276; value = partition index of sub-block b (raster) */
static const uint8_t split_layout[4][16] = {
  { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1 },
  { 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1 },
  { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 },
  { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 } };
static const uint8_t split_count[4] = { 2, 2, 4, 16 };

/* Sub-block MV/mode of the 4x4 block left of / above sub-block b of MB (col,row); outside the
 * frame -> zero MV / B_DC_PRED (macroblock.cc:237-241, :92-95). */
static const vp8o_mb * neighbour_block( const vp8o_decoder * d, int col, int row, int b, int dx, int dy, int * nb )
{
  int bx = ( b & 3 ) + dx, by = ( b >> 2 ) + dy;
  if ( bx < 0 ) { col--; bx += 4; }
  if ( by < 0 ) { row--; by += 4; }
  if ( col < 0 || row < 0 ) return NULL;
  *nb = by * 4 + bx;
  return &d->mbs[row * d->mbw + col];
}

/* ---------------- token parsing: tokens.cc:50-135 ---------------- */
static int parse_block_tokens( booldec * bd, const probtab * p, int type, int ctx, int16_t * out )
{
  static const uint8_t cat_base[5] = { 7, 11, 19, 35, 67 };
  static const uint8_t cat_len[5] = { 2, 3, 4, 5, 11 };
  static const uint8_t cat_probs[5][11] = { { 165, 145 }, { 173, 148, 140 }, { 176, 155, 140, 135 },
                                            { 180, 157, 141, 134, 130 },
                                            { 254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129 } }; /* tokens.hh:74-78 */
  int last_was_zero = 0, nonzero = 0;
  for ( int index = ( type == BT_Y_AFTER_Y2 ) ? 1 : 0; index < 16; index++ ) {
    const uint8_t * prob = p->coeff[type][coeff_band[index]][ctx];
    if ( !last_was_zero ) { if ( !bd_get( bd, prob[0] ) ) break; }    /* EOB */
    if ( !bd_get( bd, prob[1] ) ) { last_was_zero = 1; ctx = 0; continue; }
    last_was_zero = 0; nonzero = 1;                                     /* Q1 */
    int value;
    if ( !bd_get( bd, prob[2] ) ) { value = 1; ctx = 1; }
    else {
      ctx = 2;
      if ( !bd_get( bd, prob[3] ) ) {
        if ( !bd_get( bd, prob[4] ) ) value = 2;
        else value = bd_get( bd, prob[5] ) ? 4 : 3;
      } else {
        int cat;
        if ( !bd_get( bd, prob[6] ) ) {
          if ( !bd_get( bd, prob[7] ) ) { value = 5 + bd_get( bd, 159 ); cat = -1; }
          else cat = 0;
        } else {
          if ( !bd_get( bd, prob[8] ) ) cat = bd_get( bd, prob[9] ) ? 2 : 1;
          else cat = bd_get( bd, prob[10] ) ? 4 : 3;
        }
        if ( cat >= 0 ) {
          int inc = 0;
          for ( int i = 0; i < cat_len[cat]; i++ ) inc = ( inc << 1 ) + bd_get( bd, cat_probs[cat][i] );
          value = cat_base[cat] + inc;
        }
      }
    }
    if ( bd_get( bd, 128 ) ) value = -value;
    out[ zigzag[index] ] = (int16_t) value;
  }
  return nonzero;
}

/* ---------------- quantiser: quantization.cc:66-93, frame.cc:185-206 ---------------- */
typedef struct { uint16_t y_ac, y_dc, y2_ac, y2_dc, uv_ac, uv_dc; } quantizer;
static int clamp_q( int q ) { return q < 0 ? 0 : ( q > 127 ? 127 : q ); }
static quantizer make_quantizer( const frame_header * h, int y_ac_qi /* already a uint8 value (Q2) */ )
{
  quantizer q;
  q.y_ac = vp8o_ac_qlookup[ clamp_q( y_ac_qi ) ];
  q.y_dc = vp8o_dc_qlookup[ clamp_q( y_ac_qi + h->y_dc ) ];
  q.y2_ac = (uint16_t) ( vp8o_ac_qlookup[ clamp_q( y_ac_qi + h->y2_ac ) ] * 155 / 100 );
  q.y2_dc = (uint16_t) ( vp8o_dc_qlookup[ clamp_q( y_ac_qi + h->y2_dc ) ] * 2 );
  q.uv_ac = vp8o_ac_qlookup[ clamp_q( y_ac_qi + h->uv_ac ) ];
  q.uv_dc = vp8o_dc_qlookup[ clamp_q( y_ac_qi + h->uv_dc ) ];
  if ( q.y2_ac < 8 ) q.y2_ac = 8;
  if ( q.uv_dc > 132 ) q.uv_dc = 132;
  return q;
}
static void dequantize( const int16_t * in, uint16_t dc, uint16_t ac, int16_t * out ) /* quantization.cc:118-121, Q4 */
{
  out[0] = (int16_t) ( in[0] * dc );
  for ( int i = 1; i < 16; i++ ) out[i] = (int16_t) ( in[i] * ac );
}

/* ---------------- transforms: transform.cc:47-137 ---------------- */
static uint8_t clamp255( int v ) { return (uint8_t) ( v < 0 ? 0 : ( v > 255 ? 255 : v ) ); }
static int MUL_20091( int a ) { return ( ( a * 20091 ) >> 16 ) + a; }
static int MUL_35468( int a ) { return ( a * 35468 ) >> 16; }

static void idct_add( const int16_t * c, uint8_t * dst, int stride )
{
  int16_t im[16];   /* Q5: int16 intermediate */
  for ( int i = 0; i < 4; i++ ) {
    const int t0 = c[i] + c[i + 8], t1 = c[i] - c[i + 8];
    const int t2 = MUL_35468( c[i + 4] ) - MUL_20091( c[i + 12] );
    const int t3 = MUL_20091( c[i + 4] ) + MUL_35468( c[i + 12] );
    im[i * 4 + 0] = (int16_t) ( t0 + t3 ); im[i * 4 + 1] = (int16_t) ( t1 + t2 );
    im[i * 4 + 2] = (int16_t) ( t1 - t2 ); im[i * 4 + 3] = (int16_t) ( t0 - t3 );
  }
  for ( int i = 0; i < 4; i++ ) {
    const int t0 = im[i] + im[i + 8], t1 = im[i] - im[i + 8];
    const int t2 = MUL_35468( im[i + 4] ) - MUL_20091( im[i + 12] );
    const int t3 = MUL_20091( im[i + 4] ) + MUL_35468( im[i + 12] );
    uint8_t * t = dst + i * stride;
    t[0] = clamp255( t[0] + ( ( t0 + t3 + 4 ) >> 3 ) );
    t[1] = clamp255( t[1] + ( ( t1 + t2 + 4 ) >> 3 ) );
    t[2] = clamp255( t[2] + ( ( t1 - t2 + 4 ) >> 3 ) );
    t[3] = clamp255( t[3] + ( ( t0 - t3 + 4 ) >> 3 ) );
  }
}
static void iwht( const int16_t * c, int16_t ydc[16] ) /* transform.cc:55-85; ydc[row*4+col] */
{
  int16_t im[16];
  for ( int i = 0; i < 4; i++ ) {
    const int a1 = c[i] + c[i + 12], b1 = c[i + 4] + c[i + 8], c1 = c[i + 4] - c[i + 8], d1 = c[i] - c[i + 12];
    im[i] = (int16_t) ( a1 + b1 ); im[i + 4] = (int16_t) ( c1 + d1 );
    im[i + 8] = (int16_t) ( a1 - b1 ); im[i + 12] = (int16_t) ( d1 - c1 );
  }
  for ( int i = 0; i < 4; i++ ) {
    const int o = i * 4;
    const int a1 = im[o] + im[o + 3], b1 = im[o + 1] + im[o + 2], c1 = im[o + 1] - im[o + 2], d1 = im[o] - im[o + 3];
    ydc[o + 0] = (int16_t) ( ( a1 + b1 + 3 ) >> 3 ); ydc[o + 1] = (int16_t) ( ( c1 + d1 + 3 ) >> 3 );
    ydc[o + 2] = (int16_t) ( ( a1 - b1 + 3 ) >> 3 ); ydc[o + 3] = (int16_t) ( ( d1 - c1 + 3 ) >> 3 );
  }
}

/* ---------------- intra prediction: prediction.cc:99-643 ---------------- */
/* Neighbour pixels of an NxN block at pixel (x0,y0) of a plane of width pw.  above[-1..2N-1]. */
static void predictors( const uint8_t * plane, int pw, int x0, int y0, int n, uint8_t * above /* [-1..] */, uint8_t * left )
{
  for ( int i = 0; i < n; i++ ) left[i] = x0 > 0 ? plane[ ( y0 + i ) * pw + x0 - 1 ] : 129;
  for ( int i = 0; i < n; i++ ) above[i] = y0 > 0 ? plane[ ( y0 - 1 ) * pw + x0 + i ] : 127;
  if ( x0 > 0 && y0 > 0 ) above[-1] = plane[ ( y0 - 1 ) * pw + x0 - 1 ];
  else if ( y0 > 0 ) above[-1] = 129;
  else above[-1] = 127;
  if ( n != 4 ) return;
  /* above-right, 4x4 only: prediction.cc:140-164 (block units: column_=x0/4,row_=y0/4) */
  const int bc = x0 / 4, br = y0 / 4;
  const int mb_top = ( br / 4 ) * 16;   /* pixel row of the top of the macroblock */
  if ( br == 0 ) memset( above + 4, 127, 4 );
  else if ( 4 * ( bc + 1 ) >= pw ) {
    if ( br >= 4 ) memset( above + 4, plane[ ( mb_top - 1 ) * pw + 4 * ( bc + 1 ) - 1 ], 4 );
    else memset( above + 4, 127, 4 );
  } else if ( bc % 4 == 3 && br % 4 != 0 ) {
    if ( br >= 4 ) memcpy( above + 4, plane + ( mb_top - 1 ) * pw + 4 * ( bc + 1 ), 4 );
    else memset( above + 4, 127, 4 );
  } else memcpy( above + 4, plane + ( y0 - 1 ) * pw + 4 * ( bc + 1 ), 4 );
}

static void predict_big( uint8_t * plane, int pw, int x0, int y0, int n, int mode ) /* 16x16 Y / 8x8 chroma */
{
  uint8_t abuf[40], left[16]; uint8_t * above = abuf + 8;
  predictors( plane, pw, x0, y0, n, above, left );
  uint8_t * dst = plane + y0 * pw + x0;
  const int log2n = n == 16 ? 4 : 3;
  switch ( mode ) {
  case DC_PRED: {
    int v = 128;
    if ( x0 > 0 && y0 > 0 ) { int s = 0; for ( int i = 0; i < n; i++ ) s += above[i] + left[i]; v = ( s + ( 1 << log2n ) ) >> ( log2n + 1 ); }
    else if ( y0 > 0 ) { int s = 0; for ( int i = 0; i < n; i++ ) s += above[i]; v = ( s + ( 1 << ( log2n - 1 ) ) ) >> log2n; }
    else if ( x0 > 0 ) { int s = 0; for ( int i = 0; i < n; i++ ) s += left[i]; v = ( s + ( 1 << ( log2n - 1 ) ) ) >> log2n; }
    for ( int r = 0; r < n; r++ ) memset( dst + r * pw, v, n );
    break; }
  case V_PRED: for ( int r = 0; r < n; r++ ) memcpy( dst + r * pw, above, n ); break;
  case H_PRED: for ( int r = 0; r < n; r++ ) memset( dst + r * pw, left[r], n ); break;
  case TM_PRED:
    for ( int r = 0; r < n; r++ ) for ( int c = 0; c < n; c++ ) dst[r * pw + c] = clamp255( left[r] + above[c] - above[-1] );
    break;
  }
}

static uint8_t avg3( int x, int y, int z ) { return (uint8_t) ( ( x + 2 * y + z + 2 ) >> 2 ); }
static uint8_t avg2( int x, int y ) { return (uint8_t) ( ( x + y + 1 ) >> 1 ); }

static void predict_4x4( uint8_t * plane, int pw, int x0, int y0, int mode )
{
  uint8_t abuf[24], left[4]; uint8_t * A = abuf + 8;
  predictors( plane, pw, x0, y0, 4, A, left );
  uint8_t * dst = plane + y0 * pw + x0;
#define P( c, r ) dst[ ( r ) * pw + ( c ) ]
  uint8_t E[9]; /* east(i): vp8_raster.hh:79 */
  for ( int i = 0; i < 9; i++ ) E[i] = i <= 3 ? left[3 - i] : A[i - 5];
  switch ( mode ) {
  case B_DC_PRED: { int s = 4; for ( int i = 0; i < 4; i++ ) s += A[i] + left[i]; for ( int r = 0; r < 4; r++ ) memset( dst + r * pw, s >> 3, 4 ); break; }
  case B_TM_PRED: for ( int r = 0; r < 4; r++ ) for ( int c = 0; c < 4; c++ ) P( c, r ) = clamp255( left[r] + A[c] - A[-1] ); break;
  case B_VE_PRED: for ( int c = 0; c < 4; c++ ) { const uint8_t v = avg3( A[c - 1], A[c], A[c + 1] ); for ( int r = 0; r < 4; r++ ) P( c, r ) = v; } break;
  case B_HE_PRED: {
    const uint8_t v0 = avg3( A[-1], left[0], left[1] ), v1 = avg3( left[0], left[1], left[2] ),
                  v2 = avg3( left[1], left[2], left[3] ), v3 = avg3( left[2], left[3], left[3] );
    memset( dst, v0, 4 ); memset( dst + pw, v1, 4 ); memset( dst + 2 * pw, v2, 4 ); memset( dst + 3 * pw, v3, 4 ); break; }
  case B_LD_PRED:
    P( 0, 0 ) = avg3( A[0], A[1], A[2] );
    P( 1, 0 ) = P( 0, 1 ) = avg3( A[1], A[2], A[3] );
    P( 2, 0 ) = P( 1, 1 ) = P( 0, 2 ) = avg3( A[2], A[3], A[4] );
    P( 3, 0 ) = P( 2, 1 ) = P( 1, 2 ) = P( 0, 3 ) = avg3( A[3], A[4], A[5] );
    P( 3, 1 ) = P( 2, 2 ) = P( 1, 3 ) = avg3( A[4], A[5], A[6] );
    P( 3, 2 ) = P( 2, 3 ) = avg3( A[5], A[6], A[7] );
    P( 3, 3 ) = avg3( A[6], A[7], A[7] );
    break;
  case B_RD_PRED:
    P( 0, 3 ) = avg3( E[0], E[1], E[2] );
    P( 1, 3 ) = P( 0, 2 ) = avg3( E[1], E[2], E[3] );
    P( 2, 3 ) = P( 1, 2 ) = P( 0, 1 ) = avg3( E[2], E[3], E[4] );
    P( 3, 3 ) = P( 2, 2 ) = P( 1, 1 ) = P( 0, 0 ) = avg3( E[3], E[4], E[5] );
    P( 3, 2 ) = P( 2, 1 ) = P( 1, 0 ) = avg3( E[4], E[5], E[6] );
    P( 3, 1 ) = P( 2, 0 ) = avg3( E[5], E[6], E[7] );
    P( 3, 0 ) = avg3( E[6], E[7], E[8] );
    break;
  case B_VR_PRED:
    P( 0, 3 ) = avg3( E[1], E[2], E[3] );
    P( 0, 2 ) = avg3( E[2], E[3], E[4] );
    P( 1, 3 ) = P( 0, 1 ) = avg3( E[3], E[4], E[5] );
    P( 1, 2 ) = P( 0, 0 ) = avg2( E[4], E[5] );
    P( 2, 3 ) = P( 1, 1 ) = avg3( E[4], E[5], E[6] );
    P( 2, 2 ) = P( 1, 0 ) = avg2( E[5], E[6] );
    P( 3, 3 ) = P( 2, 1 ) = avg3( E[5], E[6], E[7] );
    P( 3, 2 ) = P( 2, 0 ) = avg2( E[6], E[7] );
    P( 3, 1 ) = avg3( E[6], E[7], E[8] );
    P( 3, 0 ) = avg2( E[7], E[8] );
    break;
  case B_VL_PRED:
    P( 0, 0 ) = avg2( A[0], A[1] );
    P( 0, 1 ) = avg3( A[0], A[1], A[2] );
    P( 0, 2 ) = P( 1, 0 ) = avg2( A[1], A[2] );
    P( 1, 1 ) = P( 0, 3 ) = avg3( A[1], A[2], A[3] );
    P( 1, 2 ) = P( 2, 0 ) = avg2( A[2], A[3] );
    P( 1, 3 ) = P( 2, 1 ) = avg3( A[2], A[3], A[4] );
    P( 2, 2 ) = P( 3, 0 ) = avg2( A[3], A[4] );
    P( 2, 3 ) = P( 3, 1 ) = avg3( A[3], A[4], A[5] );
    P( 3, 2 ) = avg3( A[4], A[5], A[6] );
    P( 3, 3 ) = avg3( A[5], A[6], A[7] );
    break;
  case B_HD_PRED:
    P( 0, 3 ) = avg2( E[0], E[1] );
    P( 1, 3 ) = avg3( E[0], E[1], E[2] );
    P( 0, 2 ) = P( 2, 3 ) = avg2( E[1], E[2] );
    P( 1, 2 ) = P( 3, 3 ) = avg3( E[1], E[2], E[3] );
    P( 2, 2 ) = P( 0, 1 ) = avg2( E[2], E[3] );
    P( 3, 2 ) = P( 1, 1 ) = avg3( E[2], E[3], E[4] );
    P( 2, 1 ) = P( 0, 0 ) = avg2( E[3], E[4] );
    P( 3, 1 ) = P( 1, 0 ) = avg3( E[3], E[4], E[5] );
    P( 2, 0 ) = avg3( E[4], E[5], E[6] );
    P( 3, 0 ) = avg3( E[5], E[6], E[7] );
    break;
  case B_HU_PRED:
    P( 0, 0 ) = avg2( left[0], left[1] );
    P( 1, 0 ) = avg3( left[0], left[1], left[2] );
    P( 2, 0 ) = P( 0, 1 ) = avg2( left[1], left[2] );
    P( 3, 0 ) = P( 1, 1 ) = avg3( left[1], left[2], left[3] );
    P( 2, 1 ) = P( 0, 2 ) = avg2( left[2], left[3] );
    P( 3, 1 ) = P( 1, 2 ) = avg3( left[2], left[3], left[3] );
    P( 2, 2 ) = P( 3, 2 ) = P( 0, 3 ) = P( 1, 3 ) = P( 2, 3 ) = P( 3, 3 ) = left[3];
    break;
  }
#undef P
}

/* ---------------- inter prediction: prediction.cc:655-674, 919-971; vp8_raster.hh:318-339 ----------------
 * Every fetch is coordinate-clamped to the padded plane -- identical to the unsafe path when
 * the footprint is interior, and to EdgeExtendedRaster otherwise. */
static uint8_t ref_px( const uint8_t * ref, int w, int h, int x, int y )
{ return ref[ clampi( y, 0, h - 1 ) * w + clampi( x, 0, w - 1 ) ]; }

static void inter_predict( const uint8_t * ref, int w, int h, int x0, int y0, int n, int mvx, int mvy, uint8_t * dst, int stride )
{
  const int sx = x0 + ( mvx >> 3 ), sy = y0 + ( mvy >> 3 );   /* Q9: arithmetic shift */
  const int mx = mvx & 7, my = mvy & 7;
  if ( mx == 0 && my == 0 ) {
    for ( int r = 0; r < n; r++ ) for ( int c = 0; c < n; c++ ) dst[r * stride + c] = ref_px( ref, w, h, sx + c, sy + r );
    return;
  }
  uint8_t im[21][16];
  const int16_t * hf = sixtap[mx], * vf = sixtap[my];
  for ( int r = 0; r < n + 5; r++ ) for ( int c = 0; c < n; c++ ) {
    int s = 64;
    for ( int t = 0; t < 6; t++ ) s += ref_px( ref, w, h, sx + c - 2 + t, sy + r - 2 ) * hf[t];
    im[r][c] = clamp255( s >> 7 );   /* Q6 */
  }
  for ( int r = 0; r < n; r++ ) for ( int c = 0; c < n; c++ ) {
    int s = 64;
    for ( int t = 0; t < 6; t++ ) s += im[r + t][c] * vf[t];
    dst[r * stride + c] = clamp255( s >> 7 );
  }
}

/* ---------------- loop filter: loopfilter.cc, loopfilter_filters.hh:50-183 ---------------- */
static int8_t sclamp( int t ) { return (int8_t) ( t < -128 ? -128 : ( t > 127 ? 127 : t ) ); }
static int iabs( int v ) { return v < 0 ? -v : v; }

static int filter_mask( int limit, int blimit, int p3, int p2, int p1, int p0, int q0, int q1, int q2, int q3 )
{
  int m = 0;
  m |= iabs( p3 - p2 ) > limit; m |= iabs( p2 - p1 ) > limit; m |= iabs( p1 - p0 ) > limit;
  m |= iabs( q1 - q0 ) > limit; m |= iabs( q2 - q1 ) > limit; m |= iabs( q3 - q2 ) > limit;
  m |= ( iabs( p0 - q0 ) * 2 + iabs( p1 - q1 ) / 2 ) > blimit;
  return !m;   /* 1 = filter */
}
static int hev_mask( int thresh, int p1, int p0, int q0, int q1 ) { return iabs( p1 - p0 ) > thresh || iabs( q1 - q0 ) > thresh; }

static void sb_filter( int mask, int hev, uint8_t * op1, uint8_t * op0, uint8_t * oq0, uint8_t * oq1 ) /* vp8_filter :82-130 */
{
  const int8_t ps1 = (int8_t) ( *op1 ^ 0x80 ), ps0 = (int8_t) ( *op0 ^ 0x80 ), qs0 = (int8_t) ( *oq0 ^ 0x80 ), qs1 = (int8_t) ( *oq1 ^ 0x80 );
  int8_t f = sclamp( ps1 - qs1 );
  if ( !hev ) f = 0;
  f = sclamp( f + 3 * ( qs0 - ps0 ) );
  if ( !mask ) f = 0;
  int8_t f1 = sclamp( f + 4 ), f2 = sclamp( f + 3 );
  f1 >>= 3; f2 >>= 3;
  *oq0 = (uint8_t) ( sclamp( qs0 - f1 ) ^ 0x80 );
  *op0 = (uint8_t) ( sclamp( ps0 + f2 ) ^ 0x80 );
  f = f1; f += 1; f >>= 1;
  if ( hev ) f = 0;
  *oq1 = (uint8_t) ( sclamp( qs1 - f ) ^ 0x80 );
  *op1 = (uint8_t) ( sclamp( ps1 + f ) ^ 0x80 );
}
static void mb_filter( int mask, int hev, uint8_t * op2, uint8_t * op1, uint8_t * op0, uint8_t * oq0, uint8_t * oq1, uint8_t * oq2 ) /* vp8_mbfilter :132-183 */
{
  const int8_t ps2 = (int8_t) ( *op2 ^ 0x80 ), ps1 = (int8_t) ( *op1 ^ 0x80 ); int8_t ps0 = (int8_t) ( *op0 ^ 0x80 );
  int8_t qs0 = (int8_t) ( *oq0 ^ 0x80 ); const int8_t qs1 = (int8_t) ( *oq1 ^ 0x80 ), qs2 = (int8_t) ( *oq2 ^ 0x80 );
  int8_t f = sclamp( ps1 - qs1 );
  f = sclamp( f + 3 * ( qs0 - ps0 ) );
  if ( !mask ) f = 0;
  int8_t f2 = hev ? f : 0;
  int8_t f1 = sclamp( f2 + 4 ); f2 = sclamp( f2 + 3 );
  f1 >>= 3; f2 >>= 3;
  qs0 = sclamp( qs0 - f1 ); ps0 = sclamp( ps0 + f2 );
  if ( hev ) f = 0;
  int8_t u = sclamp( ( 63 + f * 27 ) >> 7 );
  *oq0 = (uint8_t) ( sclamp( qs0 - u ) ^ 0x80 ); *op0 = (uint8_t) ( sclamp( ps0 + u ) ^ 0x80 );
  u = sclamp( ( 63 + f * 18 ) >> 7 );
  *oq1 = (uint8_t) ( sclamp( qs1 - u ) ^ 0x80 ); *op1 = (uint8_t) ( sclamp( ps1 + u ) ^ 0x80 );
  u = sclamp( ( 63 + f * 9 ) >> 7 );
  *oq2 = (uint8_t) ( sclamp( qs2 - u ) ^ 0x80 ); *op2 = (uint8_t) ( sclamp( ps2 + u ) ^ 0x80 );
}
/* one edge: `central` = first pixel on the q side, `step` = distance across the edge, `along` = distance along it */
static void filter_edge( uint8_t * central, int step, int along, int count, int is_mb_edge, int ilimit, int elimit, int hevt )
{
  for ( int i = 0; i < count; i++ ) {
    uint8_t * c = central + i * along;
    const int mask = filter_mask( ilimit, elimit, c[-4 * step], c[-3 * step], c[-2 * step], c[-step], c[0], c[step], c[2 * step], c[3 * step] );
    const int hev = hev_mask( hevt, c[-2 * step], c[-step], c[0], c[step] );
    if ( is_mb_edge ) mb_filter( mask, hev, c - 3 * step, c - 2 * step, c - step, c, c + step, c + 2 * step );
    else sb_filter( mask, hev, c - 2 * step, c - step, c, c + step );
  }
}

/* SimpleLoopFilter / NormalLoopFilter ctors (loopfilter.cc:81-125): interior limit, macroblock-edge limit, sub-block-edge limit, hev threshold */
static void filter_limits( int level, int sharpness, int key, int out[4] )
{
  int ilimit = level;
  if ( sharpness ) {
    ilimit >>= sharpness > 4 ? 2 : 1;
    if ( ilimit > 9 - sharpness ) ilimit = 9 - sharpness;
  }
  if ( ilimit < 1 ) ilimit = 1;
  int hevt = level >= 15;
  if ( level >= 40 ) hevt++;
  if ( level >= 20 && !key ) hevt++;
  out[0] = ilimit; out[1] = ( ( level + 2 ) * 2 ) + ilimit; out[2] = ( level * 2 ) + ilimit; out[3] = hevt;
}
static void loopfilter_mb( vp8o_decoder * d, int col, int row, int level, int skip_subblock_edges ) /* loopfilter.cc:81-154 */
{
  const frame_header * h = &d->hdr;
  level = level > 63 ? 63 : level;            /* clamp63; level > 0 already */
  int lim[4];
  filter_limits( level, h->sharpness, h->key, lim );
  const int ilimit = lim[0], mb_limit = lim[1], sb_limit = lim[2], hevt = lim[3];
  for ( int pass = 0; pass < 4; pass++ ) {
    /* 0: left MB edge, 1: inner vertical edges, 2: top MB edge, 3: inner horizontal edges */
    if ( pass == 0 && col == 0 ) continue;
    if ( pass == 2 && row == 0 ) continue;
    if ( ( pass == 1 || pass == 3 ) && skip_subblock_edges ) continue;
    for ( int p = 0; p < 3; p++ ) {
      const int n = p ? 8 : 16, pw = p ? d->pw / 2 : d->pw;
      uint8_t * base = d->ref[0].plane[p] + ( row * n ) * pw + col * n;
      switch ( pass ) {
      case 0: filter_edge( base, 1, pw, n, 1, ilimit, mb_limit, hevt ); break;
      case 1: for ( int e = 4; e < n; e += 4 ) filter_edge( base + e, 1, pw, n, 0, ilimit, sb_limit, hevt ); break;
      case 2: filter_edge( base, pw, 1, n, 1, ilimit, mb_limit, hevt ); break;
      case 3: for ( int e = 4; e < n; e += 4 ) filter_edge( base + e * pw, pw, 1, n, 0, ilimit, sb_limit, hevt ); break;
      }
    }
  }
}

/* ---------------- frame decode ---------------- */
static void reconstruct_mb( vp8o_decoder * d, int col, int row, const quantizer * q )
{
  const vp8o_mb * mb = &d->mbs[row * d->mbw + col];
  const int pw = d->pw, cw = d->pw / 2, ch = d->ph / 2;
  uint8_t * Y = d->ref[0].plane[0], * U = d->ref[0].plane[1], * V = d->ref[0].plane[2];
  int16_t dq[16];
  if ( mb->ref_frame == REF_CURRENT ) {      /* reconstruct_intra: macroblock.cc:523-551 */
    predict_big( U, cw, col * 8, row * 8, 8, mb->uv_mode );
    predict_big( V, cw, col * 8, row * 8, 8, mb->uv_mode );
    if ( mb->has_nonzero ) for ( int b = 0; b < 4; b++ ) {
      dequantize( mb->coeff[16 + b], q->uv_dc, q->uv_ac, dq ); idct_add( dq, U + ( row * 8 + ( b >> 1 ) * 4 ) * cw + col * 8 + ( b & 1 ) * 4, cw );
      dequantize( mb->coeff[20 + b], q->uv_dc, q->uv_ac, dq ); idct_add( dq, V + ( row * 8 + ( b >> 1 ) * 4 ) * cw + col * 8 + ( b & 1 ) * 4, cw );
    }
    if ( mb->y_mode == B_PRED ) {
      for ( int b = 0; b < 16; b++ ) {
        const int x0 = col * 16 + ( b & 3 ) * 4, y0 = row * 16 + ( b >> 2 ) * 4;
        predict_4x4( Y, pw, x0, y0, mb->b_mode[b] );
        if ( mb->has_nonzero ) { dequantize( mb->coeff[b], q->y_dc, q->y_ac, dq ); idct_add( dq, Y + y0 * pw + x0, pw ); }
      }
      return;
    }
    predict_big( Y, pw, col * 16, row * 16, 16, mb->y_mode );
  } else {                                    /* reconstruct_inter: macroblock.cc:553-601 */
    const raster * ref = &d->ref[mb->ref_frame];
    if ( mb->y_mode == SPLITMV ) {
      for ( int b = 0; b < 16; b++ ) {
        const int x0 = col * 16 + ( b & 3 ) * 4, y0 = row * 16 + ( b >> 2 ) * 4;
        inter_predict( ref->plane[0], pw, d->ph, x0, y0, 4, mb->mv[b][0], mb->mv[b][1], Y + y0 * pw + x0, pw );
      }
      for ( int b = 0; b < 4; b++ ) {
        const int x0 = col * 8 + ( b & 1 ) * 4, y0 = row * 8 + ( b >> 1 ) * 4;
        inter_predict( ref->plane[1], cw, ch, x0, y0, 4, mb->uv_mv[b][0], mb->uv_mv[b][1], U + y0 * cw + x0, cw );
        inter_predict( ref->plane[2], cw, ch, x0, y0, 4, mb->uv_mv[b][0], mb->uv_mv[b][1], V + y0 * cw + x0, cw );
      }
      if ( mb->has_nonzero ) {
        for ( int b = 0; b < 16; b++ ) { dequantize( mb->coeff[b], q->y_dc, q->y_ac, dq ); idct_add( dq, Y + ( row * 16 + ( b >> 2 ) * 4 ) * pw + col * 16 + ( b & 3 ) * 4, pw ); }
        for ( int b = 0; b < 4; b++ ) {
          dequantize( mb->coeff[16 + b], q->uv_dc, q->uv_ac, dq ); idct_add( dq, U + ( row * 8 + ( b >> 1 ) * 4 ) * cw + col * 8 + ( b & 1 ) * 4, cw );
          dequantize( mb->coeff[20 + b], q->uv_dc, q->uv_ac, dq ); idct_add( dq, V + ( row * 8 + ( b >> 1 ) * 4 ) * cw + col * 8 + ( b & 1 ) * 4, cw );
        }
      }
      return;
    }
    inter_predict( ref->plane[0], pw, d->ph, col * 16, row * 16, 16, mb->mv[15][0], mb->mv[15][1], Y + row * 16 * pw + col * 16, pw );
    inter_predict( ref->plane[1], cw, ch, col * 8, row * 8, 8, mb->uv_mv[0][0], mb->uv_mv[0][1], U + row * 8 * cw + col * 8, cw );
    inter_predict( ref->plane[2], cw, ch, col * 8, row * 8, 8, mb->uv_mv[0][0], mb->uv_mv[0][1], V + row * 8 * cw + col * 8, cw );
    if ( mb->has_nonzero ) for ( int b = 0; b < 4; b++ ) {
      dequantize( mb->coeff[16 + b], q->uv_dc, q->uv_ac, dq ); idct_add( dq, U + ( row * 8 + ( b >> 1 ) * 4 ) * cw + col * 8 + ( b & 1 ) * 4, cw );
      dequantize( mb->coeff[20 + b], q->uv_dc, q->uv_ac, dq ); idct_add( dq, V + ( row * 8 + ( b >> 1 ) * 4 ) * cw + col * 8 + ( b & 1 ) * 4, cw );
    }
  }
  /* Y2 / Walsh path: apply_walsh macroblock.cc:504-521 */
  if ( mb->has_nonzero ) {
    int16_t y2[16], ydc[16];
    dequantize( mb->coeff[24], q->y2_dc, q->y2_ac, y2 );
    iwht( y2, ydc );
    for ( int b = 0; b < 16; b++ ) {
      dequantize( mb->coeff[b], q->y_dc, q->y_ac, dq );
      dq[0] = ydc[b];
      idct_add( dq, Y + ( row * 16 + ( b >> 2 ) * 4 ) * pw + col * 16 + ( b & 3 ) * 4, pw );
    }
  }
}

static int mode_adjustment( const filteradj * f, int ref, int y_mode ) /* loopfilter.cc:59-79 */
{
  int adj = f->ref[ref];
  if ( ref == REF_CURRENT ) adj += ( y_mode == B_PRED ) ? f->mode[0] : 0;
  else if ( y_mode == ZEROMV ) adj += f->mode[1];
  else if ( y_mode == SPLITMV ) adj += f->mode[3];
  else adj += f->mode[2];
  return adj;
}

int vp8o_decode_frame( vp8o_decoder * d, const uint8_t * data, size_t size, int * shown_out )
{
  frame_header * h = &d->hdr;
  d->err[0] = 0;
  /* ---- uncompressed chunk: uncompressed_chunk.cc:34-130 ---- */
  /* accept_partial (error concealment): a first partition that reaches the end of the frame keeps what there is of it and there
   * is no DCT data (:82-95); a frame too short for its tag (the reference's Chunk throws out_of_range, :116-127) becomes an inter
   * frame with an empty first partition.  Nothing else: BoolDecoder::valid() never turns false on such a frame because
   * decoder_state.hh:79,120 passes "corrupted" as complete_chunk, so the macroblock-level branches of macroblock.cc are dead. */
  int key = 0, show = 0, experimental = 0, corrupted_frame = 0;
  uint32_t first_off = 3, first_len = 0;
  size_t rest_at = size;
  if ( size >= 1 ) {
    key = !( data[0] & 1 ); show = ( data[0] >> 4 ) & 1;
    const int version = ( data[0] >> 1 ) & 7;
    if ( version == 4 || version == 6 ) experimental = 1;
    else if ( version != 0 ) return fail( d, VP8O_UNSUPPORTED, "VP8 version" );
  }
  if ( size < 3 ) {
    if ( !d->conceal ) return fail( d, VP8O_INVALID, "VP8 frame truncated" );
    corrupted_frame = 1;
  } else {
    const uint32_t tag = data[0] | ( data[1] << 8 ) | ( (uint32_t) data[2] << 16 );
    first_len = ( tag >> 5 ) & 0x7FFFF;
    first_off = key ? 10 : 3;
    if ( size <= (size_t) first_off + first_len ) {
      if ( !d->conceal ) return fail( d, VP8O_INVALID, "invalid VP8 first partition length" );
      if ( size < first_off ) corrupted_frame = 1;
      else first_len = (uint32_t) ( size - first_off );
    } else rest_at = (size_t) first_off + first_len;
  }
  if ( corrupted_frame ) { key = 0; experimental = 0; first_off = 0; first_len = 0; rest_at = size; }
  if ( key ) {
    if ( data[3] != 0x9d || data[4] != 0x01 || data[5] != 0x2a ) return fail( d, VP8O_INVALID, "did not find key-frame start code" );
    const int fw = ( data[6] | ( data[7] << 8 ) ) & 0x3FFF, hs = data[7] >> 6;
    const int fh = ( data[8] | ( data[9] << 8 ) ) & 0x3FFF, vs = data[9] >> 6;
    if ( fw != d->width || fh != d->height || hs || vs ) return fail( d, VP8O_UNSUPPORTED, "VP8 upscaling not supported" );
    if ( experimental ) return fail( d, VP8O_INVALID, "experimental key frame" );     /* decoder_state.hh:81-83 */
  } else if ( experimental ) return fail( d, VP8O_UNSUPPORTED, "experimental" );      /* decoder.cc:131-133 */
  const uint8_t * first = data + first_off;
  const uint8_t * rest = data + rest_at; size_t rest_len = size - rest_at;

  booldec bd; bd_init( &bd, first, first_len );
  memset( h, 0, sizeof *h );
  h->key = key; h->show = show;
  probtab fp;   /* this frame's probability tables */

  /* ---- frame header + state transition: decoder_state.hh:72-167 ---- */
  if ( key ) {
    const int color_space = bd_flag( &bd ), clamping_type = bd_flag( &bd );
    parse_segmentation_and_filter( &bd, h );
    h->refresh_entropy = bd_flag( &bd );
    probs_default( &d->probs );                       /* key frame resets persistent state: decoder.cc:234-240 */
    fp = d->probs;
    parse_token_prob_updates( &bd, &fp );
    h->skip_enabled = bd_flag( &bd ); h->prob_skip = h->skip_enabled ? bd_uint( &bd, 8 ) : 0;
    if ( color_space || clamping_type ) return fail( d, VP8O_UNSUPPORTED, "VP8 color_space and clamping_type bits" );
    if ( h->filter_type ) return fail( d, VP8O_UNSUPPORTED, "VP8 'simple' in-loop deblocking filter" );
    h->refresh_last = h->refresh_golden = h->refresh_alt = 1;
    /* DecoderState(KeyFrameHeader): fresh Segmentation / FilterAdjustments iff present in the header */
    d->seg.enabled = h->seg_enabled; d->seg.absolute = 0;
    memset( d->seg.quant, 0, 4 ); memset( d->seg.lf, 0, 4 );
    d->fadj.enabled = h->lf_adj_enabled; memset( d->fadj.ref, 0, 4 ); memset( d->fadj.mode, 0, 4 );
    if ( h->seg_enabled ) memset( d->seg.map, 3, (size_t) d->mbw * d->mbh );   /* Segmentation ctor: map(width,height,3) */
  } else {
    parse_segmentation_and_filter( &bd, h );
    h->refresh_golden = bd_flag( &bd ); h->refresh_alt = bd_flag( &bd );
    h->copy_golden = h->refresh_golden ? 0 : bd_uint( &bd, 2 );
    h->copy_alt = h->refresh_alt ? 0 : bd_uint( &bd, 2 );
    h->sign_bias_golden = bd_flag( &bd ); h->sign_bias_alt = bd_flag( &bd );
    h->refresh_entropy = bd_flag( &bd ); h->refresh_last = bd_flag( &bd );
    fp = d->probs;
    parse_token_prob_updates( &bd, &fp );
    h->skip_enabled = bd_flag( &bd ); h->prob_skip = h->skip_enabled ? bd_uint( &bd, 8 ) : 0;
    h->prob_inter = bd_uint( &bd, 8 ); h->prob_last = bd_uint( &bd, 8 ); h->prob_golden = bd_uint( &bd, 8 );
    if ( bd_flag( &bd ) ) for ( int i = 0; i < 4; i++ ) fp.y_mode[i] = (uint8_t) bd_uint( &bd, 8 );
    if ( bd_flag( &bd ) ) for ( int i = 0; i < 3; i++ ) fp.uv_mode[i] = (uint8_t) bd_uint( &bd, 8 );
    for ( int i = 0; i < 2; i++ ) for ( int j = 0; j < 19; j++ )
      if ( bd_get( &bd, vp8o_mv_update_probs[i * 19 + j] ) ) { const int x = bd_uint( &bd, 7 ); fp.mv[i][j] = (uint8_t) ( x ? x << 1 : 1 ); }
    if ( h->filter_type ) return fail( d, VP8O_UNSUPPORTED, "VP8 'simple' in-loop deblocking filter" );
    /* filter adjustments: decoder_state.hh:133-142 */
    if ( h->lf_adj_enabled ) { if ( !d->fadj.enabled ) { d->fadj.enabled = 1; memset( d->fadj.ref, 0, 4 ); memset( d->fadj.mode, 0, 4 ); } }
    else d->fadj.enabled = 0;
    /* segmentation: :144-153 */
    if ( h->seg_enabled ) {
      if ( !d->seg.enabled ) { d->seg.enabled = 1; d->seg.absolute = 0; memset( d->seg.quant, 0, 4 ); memset( d->seg.lf, 0, 4 );
                               memset( d->seg.map, 3, (size_t) d->mbw * d->mbh ); }
    } else d->seg.enabled = 0;
  }
  if ( h->refresh_entropy ) d->probs = fp;
  if ( h->lf_adj_enabled && h->lf_delta_update )
    for ( int i = 0; i < 4; i++ ) { d->fadj.ref[i] = (int8_t) h->ref_delta[i]; d->fadj.mode[i] = (int8_t) h->mode_delta[i]; }
  if ( h->seg_enabled && h->seg_update_data ) {
    d->seg.absolute = h->seg_abs;
    for ( int i = 0; i < 4; i++ ) { d->seg.quant[i] = (int8_t) h->seg_quant[i]; d->seg.lf[i] = (int8_t) h->seg_lf[i]; }
  }

  /* ---- DCT partitions: uncompressed_chunk.cc:132-155 ---- */
  const int nparts = 1 << h->log2_parts;
  booldec parts[8];
  {
    if ( rest_len < (size_t) 3 * ( nparts - 1 ) ) return fail( d, VP8O_INVALID, "partition table truncated" );
    const uint8_t * p = rest + 3 * ( nparts - 1 ); size_t left = rest_len - 3 * ( nparts - 1 );
    for ( int i = 0; i < nparts; i++ ) {
      size_t len = left;
      if ( i < nparts - 1 ) {
        len = rest[3 * i] | ( rest[3 * i + 1] << 8 ) | ( (size_t) rest[3 * i + 2] << 16 );
        if ( len > left ) return fail( d, VP8O_INVALID, "partition truncated" );
      }
      bd_init( &parts[i], p, len ); p += len; left -= len;
    }
  }

  /* ---- macroblock headers (first partition) + tokens: frame.cc:95-137, macroblock.cc ---- */
  const int mbw = d->mbw, mbh = d->mbh;
  uint8_t * flipped = (uint8_t *) calloc( (size_t) mbw * mbh, 1 );
  memset( d->above_nz, 0, (size_t) mbw * 9 );
  for ( int row = 0; row < mbh; row++ ) {
    uint8_t left_nz[9]; memset( left_nz, 0, 9 );
    for ( int col = 0; col < mbw; col++ ) {
      vp8o_mb * mb = &d->mbs[row * mbw + col];
      memset( mb, 0, sizeof *mb );
      /* Macroblock ctor: macroblock.cc:43-71 */
      if ( h->seg_enabled && h->seg_update_map ) d->seg.map[row * mbw + col] = (uint8_t) bd_tree( &bd, segment_id_tree, h->seg_tree_probs );
      mb->segment_id = d->seg.enabled ? d->seg.map[row * mbw + col] : 0;
      mb->skip = h->skip_enabled ? (uint8_t) bd_get( &bd, h->prob_skip ) : 0;
      int is_inter = 0;
      if ( !key ) {
        is_inter = bd_get( &bd, h->prob_inter );
        if ( is_inter ) {
          mb->ref_frame = REF_LAST;
          if ( bd_get( &bd, h->prob_last ) ) mb->ref_frame = bd_get( &bd, h->prob_golden ) ? REF_ALT : REF_GOLDEN;
          flipped[row * mbw + col] = ( mb->ref_frame == REF_GOLDEN && h->sign_bias_golden ) || ( mb->ref_frame == REF_ALT && h->sign_bias_alt );
        }
      }
      if ( !is_inter ) {
        /* decode_prediction_modes: key :84-111, inter-frame intra MB :354-376 */
        mb->y_mode = key ? (uint8_t) bd_tree( &bd, kf_y_mode_tree, vp8o_kf_y_mode_probs ) : (uint8_t) bd_tree( &bd, y_mode_tree, fp.y_mode );
        for ( int b = 0; b < 16; b++ ) {
          if ( mb->y_mode == B_PRED ) {
            if ( key ) {
              int nb; const vp8o_mb * am = neighbour_block( d, col, row, b, 0, -1, &nb ); const int above_mode = am ? am->b_mode[nb] : B_DC_PRED;
              const vp8o_mb * lm = neighbour_block( d, col, row, b, -1, 0, &nb ); const int left_mode = lm ? lm->b_mode[nb] : B_DC_PRED;
              mb->b_mode[b] = (uint8_t) bd_tree( &bd, b_mode_tree, vp8o_kf_b_mode_probs + ( above_mode * 10 + left_mode ) * 9 );
            } else mb->b_mode[b] = (uint8_t) bd_tree( &bd, b_mode_tree, vp8o_b_mode_probs );
          } else {
            static const uint8_t implied[4] = { B_DC_PRED, B_VE_PRED, B_HE_PRED, B_TM_PRED };   /* macroblock.hh:134-143 */
            mb->b_mode[b] = implied[mb->y_mode];
          }
        }
        mb->uv_mode = key ? (uint8_t) bd_tree( &bd, uv_mode_tree, vp8o_kf_uv_mode_probs ) : (uint8_t) bd_tree( &bd, uv_mode_tree, fp.uv_mode );
      } else {
        /* inter MB: macroblock.cc:377-455 */
        census cs; census_run( d, col, row, flipped[row * mbw + col], flipped, &cs );
        uint8_t mv_ref_probs[4];
        for ( int i = 0; i < 4; i++ ) mv_ref_probs[i] = vp8o_mv_counts_to_probs[ cs.ctx[i] * 4 + i ];
        mb->y_mode = (uint8_t) bd_tree( &bd, mv_ref_tree, mv_ref_probs );
        mvec base; base.x = base.y = 0;
        switch ( mb->y_mode ) {
        case NEARESTMV: base = mv_clamp( cs.nearest, col, row, mbw, mbh ); break;
        case NEARMV: base = mv_clamp( cs.near, col, row, mbw, mbh ); break;
        case ZEROMV: break;
        case NEWMV: { mvec nm = mv_read( &bd, &fp ); const mvec b = mv_clamp( cs.best, col, row, mbw, mbh );
                      base.x = (int16_t) ( nm.x + b.x ); base.y = (int16_t) ( nm.y + b.y ); break; }
        case SPLITMV: {
          mb->split_partition = (uint8_t) bd_tree( &bd, split_mv_tree, vp8o_split_mv_probs );
          const mvec best = mv_clamp( cs.best, col, row, mbw, mbh );
          const uint8_t * layout = split_layout[mb->split_partition];
          for ( int part = 0; part < split_count[mb->split_partition]; part++ ) {
            int first = 0; while ( layout[first] != part ) first++;
            /* read_subblock_inter_prediction: macroblock.cc:231-281 */
            int nb; mvec lmv = { 0, 0 }, amv = { 0, 0 };
            const vp8o_mb * lm = neighbour_block( d, col, row, first, -1, 0, &nb ); if ( lm ) { lmv.x = lm->mv[nb][0]; lmv.y = lm->mv[nb][1]; }
            const vp8o_mb * am = neighbour_block( d, col, row, first, 0, -1, &nb ); if ( am ) { amv.x = am->mv[nb][0]; amv.y = am->mv[nb][1]; }
            const int lz = lmv.x == 0 && lmv.y == 0, az = amv.x == 0 && amv.y == 0, eq = lmv.x == amv.x && lmv.y == amv.y;
            int ctx = 0;
            if ( eq && lz ) ctx = 4; else if ( eq ) ctx = 3; else if ( az ) ctx = 2; else if ( lz ) ctx = 1;
            const int sub_mode = bd_tree( &bd, submv_ref_tree, vp8o_submv_ref_probs + ctx * 3 );
            mvec m = { 0, 0 };
            switch ( sub_mode ) {
            case LEFT4X4: m = lmv; break;
            case ABOVE4X4: m = amv; break;
            case ZERO4X4: break;
            case NEW4X4: { const mvec nm = mv_read( &bd, &fp ); m.x = (int16_t) ( nm.x + best.x ); m.y = (int16_t) ( nm.y + best.y ); break; }
            }
            for ( int b = 0; b < 16; b++ ) if ( layout[b] == part ) { mb->mv[b][0] = m.x; mb->mv[b][1] = m.y; mb->b_mode[b] = (uint8_t) sub_mode; }
          }
          break; }
        }
        if ( mb->y_mode != SPLITMV ) for ( int b = 0; b < 16; b++ ) { mb->mv[b][0] = base.x; mb->mv[b][1] = base.y; }
        for ( int b = 0; b < 4; b++ ) {     /* chroma MVs: macroblock.cc:443-454 */
          const int bx = ( b & 1 ) * 2, by = ( b >> 1 ) * 2;
          const int i0 = by * 4 + bx, i1 = i0 + 1, i2 = i0 + 4, i3 = i0 + 5;
          mb->uv_mv[b][0] = chroma_round( (int16_t) ( mb->mv[i0][0] + mb->mv[i1][0] + mb->mv[i2][0] + mb->mv[i3][0] ) );
          mb->uv_mv[b][1] = chroma_round( (int16_t) ( mb->mv[i0][1] + mb->mv[i1][1] + mb->mv[i2][1] + mb->mv[i3][1] ) );
        }
      }
      mb->has_y2 = !( mb->y_mode == B_PRED || mb->y_mode == SPLITMV );   /* block.hh:183-190 */

      /* tokens: Macroblock::parse_tokens macroblock.cc:475-502; contexts = has_nonzero of the
       * above/left block, Y2 neighbours skip MBs without Y2 (frame.cc:255-269) */
      uint8_t * anz = d->above_nz + col * 9;
      if ( mb->skip ) {
        /* nothing parsed: every block of this MB has has_nonzero=false; a non-coded Y2 leaves
         * the Y2 context chain untouched */
        memset( anz, 0, 8 ); memset( left_nz, 0, 8 );
        if ( mb->has_y2 ) { anz[8] = 0; left_nz[8] = 0; }
      } else {
        booldec * tp = &parts[row % nparts];
        if ( mb->has_y2 ) {
          const int nz = parse_block_tokens( tp, &fp, BT_Y2, anz[8] + left_nz[8], mb->coeff[24] );
          anz[8] = left_nz[8] = (uint8_t) nz; mb->block_nonzero[24] = (uint8_t) nz; mb->has_nonzero |= nz;
        }
        const int ytype = mb->has_y2 ? BT_Y_AFTER_Y2 : BT_Y_NO_Y2;
        for ( int b = 0; b < 16; b++ ) {
          const int bx = b & 3, by = b >> 2;
          const int nz = parse_block_tokens( tp, &fp, ytype, anz[bx] + left_nz[by], mb->coeff[b] );
          anz[bx] = left_nz[by] = (uint8_t) nz; mb->block_nonzero[b] = (uint8_t) nz; mb->has_nonzero |= nz;
        }
        for ( int pl = 0; pl < 2; pl++ ) for ( int b = 0; b < 4; b++ ) {
          const int bx = b & 1, by = b >> 1;
          const int nz = parse_block_tokens( tp, &fp, BT_UV, anz[4 + pl * 2 + bx] + left_nz[4 + pl * 2 + by], mb->coeff[16 + pl * 4 + b] );
          anz[4 + pl * 2 + bx] = left_nz[4 + pl * 2 + by] = (uint8_t) nz; mb->block_nonzero[16 + pl * 4 + b] = (uint8_t) nz; mb->has_nonzero |= nz;
        }
      }
    }
  }
  free( flipped );

  /* ---- reconstruct: Frame::decode frame.cc:208-250 ---- */
  quantizer fq = make_quantizer( h, h->y_ac_qi ), sq[4];
  for ( int i = 0; i < 4; i++ ) {
    const uint8_t qi = (uint8_t) ( d->seg.quant[i] + ( d->seg.absolute ? 0 : h->y_ac_qi ) );   /* Q2: wraps as uint8 */
    sq[i] = make_quantizer( h, qi );
  }
  if ( d->phases & 1 )
    for ( int row = 0; row < mbh; row++ ) for ( int col = 0; col < mbw; col++ ) {
      const vp8o_mb * mb = &d->mbs[row * mbw + col];
      reconstruct_mb( d, col, row, d->seg.enabled ? &sq[mb->segment_id] : &fq );
    }

  /* ---- loop filter: Frame::loopfilter frame.cc:139-182, Macroblock::loopfilter macroblock.cc:603-641 ---- */
  if ( h->lf_level && ( d->phases & 2 ) ) {
    for ( int row = 0; row < mbh; row++ ) for ( int col = 0; col < mbw; col++ ) {
      const vp8o_mb * mb = &d->mbs[row * mbw + col];
      int level = h->lf_level;
      if ( d->seg.enabled ) level = d->seg.lf[mb->segment_id] + ( d->seg.absolute ? 0 : h->lf_level );   /* Q3: no clamp */
      if ( d->fadj.enabled ) level += mode_adjustment( &d->fadj, mb->ref_frame, mb->y_mode );
      if ( level <= 0 ) continue;
      loopfilter_mb( d, col, row, level, mb->has_y2 && !mb->has_nonzero );
    }
  }

  /* ---- reference update: Frame::copy_to frame.cc:271-307 ---- */
  if ( key ) { raster_copy( d, REF_LAST, 0 ); raster_copy( d, REF_GOLDEN, 0 ); raster_copy( d, REF_ALT, 0 ); }
  else {
    if ( h->copy_alt == 1 ) raster_copy( d, REF_ALT, REF_LAST ); else if ( h->copy_alt == 2 ) raster_copy( d, REF_ALT, REF_GOLDEN );
    if ( h->copy_golden == 1 ) raster_copy( d, REF_GOLDEN, REF_LAST ); else if ( h->copy_golden == 2 ) raster_copy( d, REF_GOLDEN, REF_ALT );
    if ( h->refresh_golden ) raster_copy( d, REF_GOLDEN, 0 );
    if ( h->refresh_alt ) raster_copy( d, REF_ALT, 0 );
    if ( h->refresh_last ) raster_copy( d, REF_LAST, 0 );
  }
  if ( shown_out ) *shown_out = show;
  return VP8O_OK;
}

/* ---------------- single stages ----------------
 * The arithmetic stages of the path one at a time, on caller-supplied numbers: what tests/test_gpu_stages.py compares the
 * product's device functions with, stage by stage, so that a raster mismatch can be traced to a stage and not just to a
 * macroblock.  Thin wrappers around the functions above (no restatement of their own). */
void vp8o_stage_residual( const int16_t coeff[16], int dc_q, int ac_q, uint8_t pixels[16] ) /* dequantize + idct_add onto a 4x4 prediction */
{
  int16_t dq[16];
  dequantize( coeff, (uint16_t) dc_q, (uint16_t) ac_q, dq );
  idct_add( dq, pixels, 4 );
}
void vp8o_stage_iwht( const int16_t coeff[16], int dc_q, int ac_q, int16_t ydc[16] )          /* dequantize + iwht of a Y2 block */
{
  int16_t dq[16];
  dequantize( coeff, (uint16_t) dc_q, (uint16_t) ac_q, dq );
  iwht( dq, ydc );
}
/* intra prediction of the n x n block at (x0, y0) of `plane` (width pw), in place: n = 4 -> a sub-block mode (B_DC_PRED..), else a 16x16 / 8x8 mode */
void vp8o_stage_predict( uint8_t * plane, int pw, int x0, int y0, int n, int mode )
{
  if ( n == 4 ) predict_4x4( plane, pw, x0, y0, mode );
  else predict_big( plane, pw, x0, y0, n, mode );
}
/* one six-tap pass over six pixels (prediction.cc:645-653, 875-881): clamp255( ( sum + 64 ) >> 7 ) */
int vp8o_stage_sixtap( const uint8_t p[6], int frac )
{
  int s = 64;
  for ( int t = 0; t < 6; t++ ) s += p[t] * sixtap[frac][t];
  return clamp255( s >> 7 );
}
/* both passes of an n x n block (inter_predict above) */
void vp8o_stage_inter_predict( const uint8_t * ref, int w, int h, int x0, int y0, int n, int mvx, int mvy, uint8_t * dst, int stride )
{
  inter_predict( ref, w, h, x0, y0, n, mvx, mvy, dst, stride );
}
/* one position of one loop-filter edge: px[0..7] = p3 p2 p1 p0 q0 q1 q2 q3, in place */
void vp8o_stage_filter_edge( uint8_t px[8], int is_mb_edge, int interior_limit, int edge_limit, int hev_threshold )
{
  filter_edge( px + 4, 1, 0, 1, is_mb_edge, interior_limit, edge_limit, hev_threshold );
}
/* the limits NormalLoopFilter derives from a level (loopfilter.cc:81-125): out = interior, mb edge, sub-block edge, hev threshold */
void vp8o_stage_filter_limits( int level, int sharpness, int key_frame, int out[4] ) { filter_limits( level, sharpness, key_frame, out ); }
