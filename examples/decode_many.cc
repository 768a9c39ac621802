// decode-many: N independent IVF files (any frame sizes) decoded in LOCK STEP -- frame k of every file is one GPU batch
// step (Decoder::get_frame_outputs -> aa_decode_batch) -- each written as YUV4MPEG2 to <input>.y4m (or DIR/<basename>.y4m
// with -d DIR).  This is the shape ExCamera-style bulk decoding (frontend/decode-bundle.cc:56-99 loops over its chunks
// one decoder at a time) takes on a GPU: the chip is filled by the number of streams, not by one frame.
//   g++ -std=c++14 -O2 -Iinclude examples/decode_many.cc -Lalfalfa_amd/lib -lalfalfa_amd -Wl,-rpath,$PWD/alfalfa_amd/lib
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <getopt.h>
#include <iostream>

using namespace std;

int main( int argc, char * argv[] )
{
  try {
    string outdir;
    while ( true ) {
      const int opt = getopt( argc, argv, "d:" );
      if ( opt == -1 ) break;
      if ( opt == 'd' ) outdir = optarg; else { cerr << "Usage: " << argv[0] << " [-d output_dir] input.ivf...\n"; return EXIT_FAILURE; }
    }
    if ( optind >= argc ) { cerr << "Usage: " << argv[0] << " [-d output_dir] input.ivf...\n"; return EXIT_FAILURE; }

    vector<IVF> files;
    vector<unique_ptr<FramePlayer>> players;
    vector<FileDescriptor> outputs;
    vector<unsigned int> next;                      // next frame of each file
    for ( int i = optind; i < argc; i++ ) {
      files.emplace_back( argv[i] );
      if ( files.back().fourcc() != "VP80" ) throw Unsupported( "not a VP8 file" );
      players.emplace_back( new FramePlayer( files.back().width(), files.back().height() ) );
      string name = argv[i];
      if ( not outdir.empty() ) { const size_t slash = name.find_last_of( '/' ); name = outdir + "/" + ( slash == string::npos ? name : name.substr( slash + 1 ) ); }
      outputs.emplace_back( fopen( ( name + ".y4m" ).c_str(), "wb" ) );
      unsigned int first = 0;                       // start at the first key frame, like FilePlayer (player.cc:96-105)
      while ( first < files.back().frame_count() and ( files.back().frame( first ).octet() & 1 ) ) first++;
      next.push_back( first );
    }

    while ( true ) {
      vector<Decoder *> decoders; vector<Chunk> frames; vector<size_t> who;
      for ( size_t i = 0; i < files.size(); i++ ) {
        if ( next[i] >= files[i].frame_count() ) continue;
        decoders.push_back( &players[i]->mutable_decoder() ); frames.push_back( files[i].frame( next[i]++ ) ); who.push_back( i );
      }
      if ( decoders.empty() ) break;
      const auto out = Decoder::get_frame_outputs( decoders, frames );
      for ( size_t k = 0; k < out.size(); k++ ) {
        if ( not out[k].first ) continue;           // hidden frame: references updated, nothing shown
        FileDescriptor & fd = outputs[who[k]];
        if ( fd.tell() == 0 ) fd.write( YUV4MPEGHeader( out[k].second ).to_string() );
        YUV4MPEGFrameWriter::write( out[k].second, fd );
      }
    }
  } catch ( const exception & e ) {
    print_exception( argv[0], e );
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
