/* oracle/_ref tool: prints the reference's constant VP8 tables (RFC 6386 constants as the
 * reference compiles them: vp8_prob_data.cc, modemv_data.cc, quantization.cc:42-64) in a
 * neutral "name n v0 v1 ..." text form.  tools/gen_tables.py turns that into our own
 * flat C header; tests re-run the comparison so the committed header stays pinned. */
#include <cstdio>
#include "vp8_prob_data.hh"
#include "modemv_data.hh"
#include "frame_header.hh"
#include "quantization.hh"

static void head( const char * name, unsigned n ) { printf( "%s %u", name, n ); }

int main()
{
  head( "default_coeff_probs", 4 * 8 * 3 * 11 );
  for ( unsigned i = 0; i < 4; i++ ) for ( unsigned j = 0; j < 8; j++ ) for ( unsigned k = 0; k < 3; k++ )
    for ( unsigned l = 0; l < 11; l++ ) printf( " %u", k_default_coeff_probs.at( i ).at( j ).at( k ).at( l ) );
  printf( "\n" );
  head( "coeff_update_probs", 4 * 8 * 3 * 11 );
  for ( unsigned i = 0; i < 4; i++ ) for ( unsigned j = 0; j < 8; j++ ) for ( unsigned k = 0; k < 3; k++ )
    for ( unsigned l = 0; l < 11; l++ ) printf( " %u", k_coeff_entropy_update_probs.at( i ).at( j ).at( k ).at( l ) );
  printf( "\n" );
  head( "default_y_mode_probs", 4 ); for ( unsigned i = 0; i < 4; i++ ) printf( " %u", k_default_y_mode_probs.at( i ) ); printf( "\n" );
  head( "default_uv_mode_probs", 3 ); for ( unsigned i = 0; i < 3; i++ ) printf( " %u", k_default_uv_mode_probs.at( i ) ); printf( "\n" );
  head( "mv_update_probs", 2 * 19 ); for ( unsigned i = 0; i < 2; i++ ) for ( unsigned j = 0; j < 19; j++ ) printf( " %u", k_mv_entropy_update_probs.at( i ).at( j ) ); printf( "\n" );
  head( "default_mv_probs", 2 * 19 ); for ( unsigned i = 0; i < 2; i++ ) for ( unsigned j = 0; j < 19; j++ ) printf( " %u", k_default_mv_probs.at( i ).at( j ) ); printf( "\n" );
  head( "kf_y_mode_probs", 4 ); for ( unsigned i = 0; i < 4; i++ ) printf( " %u", kf_y_mode_probs.at( i ) ); printf( "\n" );
  head( "kf_uv_mode_probs", 3 ); for ( unsigned i = 0; i < 3; i++ ) printf( " %u", kf_uv_mode_probs.at( i ) ); printf( "\n" );
  head( "kf_b_mode_probs", 10 * 10 * 9 );
  for ( unsigned i = 0; i < 10; i++ ) for ( unsigned j = 0; j < 10; j++ ) for ( unsigned k = 0; k < 9; k++ )
    printf( " %u", kf_b_mode_probs.at( i ).at( j ).at( k ) );
  printf( "\n" );
  head( "b_mode_probs", 9 ); for ( unsigned i = 0; i < 9; i++ ) printf( " %u", invariant_b_mode_probs.at( i ) ); printf( "\n" );
  head( "mv_counts_to_probs", 6 * 4 ); for ( unsigned i = 0; i < 6; i++ ) for ( unsigned j = 0; j < 4; j++ ) printf( " %u", mv_counts_to_probs.at( i ).at( j ) ); printf( "\n" );
  head( "split_mv_probs", 3 ); for ( unsigned i = 0; i < 3; i++ ) printf( " %u", split_mv_probs.at( i ) ); printf( "\n" );
  head( "submv_ref_probs", 5 * 3 ); for ( unsigned i = 0; i < 5; i++ ) for ( unsigned j = 0; j < 3; j++ ) printf( " %u", submv_ref_probs2.at( i ).at( j ) ); printf( "\n" );
  /* quantiser lookups are file-static in quantization.cc; recover them through Quantizer */
  head( "ac_qlookup", 128 );
  for ( unsigned i = 0; i < 128; i++ ) { QuantIndices qi; qi.y_ac_qi = i; Quantizer q( qi ); printf( " %u", q.y_ac ); }
  printf( "\n" );
  head( "dc_qlookup", 128 );
  for ( unsigned i = 0; i < 128; i++ ) { QuantIndices qi; qi.y_ac_qi = i; Quantizer q( qi ); printf( " %u", q.y_dc ); }
  printf( "\n" );
  return 0;
}
