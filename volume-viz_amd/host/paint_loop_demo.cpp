// paint_loop_demo.cpp -- drives PaintLoop the way a user drives the reference's 3D view and dumps what it shows:
//   paint_loop_demo <out_dir> <widget_w> <widget_h> [volume.t3d]
// frame0: first paint; (second paint without a change: nothing is marched); frame1: after an orbit drag + Phong;
// frame2: with a coronal cross-section; each with its two first-pass images.  tests/test_host_mirror.py renders the same
// three frames with the oracle from those images.  With a .t3d path: frame3 after GLWidget::loadVolume(path) (camera reset, table and
// scale by the file name, cutting plane dropped).
#include "paint_loop.h"
#include <cstdio>
#include <cstdlib>
#include <string>

static void dump(const std::string &p, const std::vector<unsigned char> &v)
{
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", p.c_str()); exit(2); }
    fwrite(v.data(), 1, v.size(), f);
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc != 4 && argc != 5) return 2;
    const std::string out = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    initCuda();                                                     // glwidget.cpp:182
    VolumeGenerator gen(64, 64, 64);
    gen.drawDefaultBrain();                                         // loadVolume(DEFAULT...) stand-in, :183
    size_t size;
    byte *texels = gen.getBytes(size);
    float tf[1024];
    vv_transfer_preset(VV_TF_ENGINE, tf);
    PaintLoop gl;
    cudaLoadVolume(texels, size, gen.getDims(), tf, &gl.volumeArray);
    gl.resizeGL(w, h);
    int marched = 0;
    marched += gl.paintGL();                                        // dirty after resize: marches
    dump(out + "/frame0.rgba", gl.resultTexture()); dump(out + "/front0.rgba", gl.frontFace()); dump(out + "/back0.rgba", gl.backFace());
    const float t0 = gl.lastRenderTime();
    marched += gl.paintGL();                                        // nothing changed: the texture is re-shown, runCuda is not called
    if (gl.runCudaCalls() != 1 || gl.lastRenderTime() != t0) { fprintf(stderr, "a clean frame was marched again\n"); return 1; }
    gl.orbitDrag(37, -21); gl.setPhongShading(true);
    marched += gl.paintGL();
    dump(out + "/frame1.rgba", gl.resultTexture()); dump(out + "/front1.rgba", gl.frontFace()); dump(out + "/back1.rgba", gl.backFace());
    gl.setSliceCanonical(CORONAL, 0.3f);                            // no slice visualisation yet: ignored (glwidget.cpp:761)
    if (gl.hasCuttingPlane() || gl.renderingDirty()) { fprintf(stderr, "setSliceCanonical took effect without a slice visualisation\n"); return 1; }
    gl.setSliceVisualization(2); gl.setSliceCanonical(CORONAL, 0.1f); gl.zoom(60);
    marched += gl.paintGL();
    dump(out + "/frame2.rgba", gl.resultTexture()); dump(out + "/front2.rgba", gl.frontFace()); dump(out + "/back2.rgba", gl.backFace());
    gl.setSlicePro(0.05f, -0.1f, 0.02f, 0.4f, -0.7f, 1.1f);         // the free-form slice sliders (window.cpp:405-443 -> setSlicePro)
    marched += gl.paintGL();
    dump(out + "/frame4.rgba", gl.resultTexture()); dump(out + "/front4.rgba", gl.frontFace()); dump(out + "/back4.rgba", gl.backFace());
    int expected = 4;
    if (argc == 5) {
        gl.loadVolume(argv[4]);                                     // window.cpp:347-351 -> glwidget.cpp:668-710
        marched += gl.paintGL(); ++expected;
        dump(out + "/frame3.rgba", gl.resultTexture()); dump(out + "/front3.rgba", gl.frontFace()); dump(out + "/back3.rgba", gl.backFace());
        printf("loadVolume: preset %d scale %g %g %g\n", gl.transferPreset(), (double)gl.scale()[0], (double)gl.scale()[1], (double)gl.scale()[2]);
    }
    printf("paint_loop_demo ok: %d x %d widget, render %d x %d, %d of %d paints marched, last render time %.4f s\n",
           gl.width(), gl.height(), gl.renderWidth(), gl.renderHeight(), marched, expected + 1, (double)gl.lastRenderTime());
    return marched == expected ? 0 : 1;
}
