// ref_driver.cc -- deterministic single-threaded driver around the REFERENCE's own
// preamble / slicer / CRC code, compiled by path from /root/reference/lib (never copied)
// against oracle/gr_stub.  Output: oracle/_ref/libairmodes_ref.so (git-ignored).
//
// TEST INFRASTRUCTURE ONLY.  Used to (1) pin oracle/airmodes_oracle.c against the real
// reference code, (2) generate tests/golden fixtures, (3) optionally time the reference's
// preamble+slicer on the GPU box's host as a reported CPU baseline.
//
// "Scheduler": general_work() is called repeatedly on the whole remaining stream (the
// reference emits at most one burst per call, preamble_impl.cc:234-238); the slicer's
// work() is called once on the concatenated bursts.  No threads, no circular buffers.
#include <gr_air_modes/types.h>
#include <gr_air_modes/modes_crc.h>
#include <gr_air_modes/preamble.h>
#include <gr_air_modes/slicer.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {

// CRC of the first nbytes bytes: straight call into modes_crc.cc:55.
uint32_t ref_crc24(const uint8_t *data, int nbytes)
{
    return modes_check_crc(const_cast<unsigned char *>(data), nbytes);
}

// Run the reference preamble block over (in, inavg) [n floats each, no history prepended]
// and the reference slicer over what it emits.
//   bursts      : cap_tags*240 floats (reference preamble output)
//   tag_sample  : item count reconstructed from the tag timestamp (secs*rate + frac*rate)
//   msgs        : '\n'-separated message texts exactly as posted to the msg_queue
// pad_items zeros are appended so that the end-of-buffer rule never fires inside the
// real data; the caller applies the canonical end-of-stream rule.
// Returns 0 on success.
//
// rx_time tags (n_tt entries, ascending item offsets in the preamble block's own item count): the
// reference latches the newest tag inside the window of the current general_work() call
// (preamble_impl.cc:165-170), so WHEN a tag takes effect depends on the scheduler's windows.  This
// driver ends every window at the next tag's offset, so that a tag is in force exactly from its
// offset on; the few items in front of a tag that the block cannot scan in such a window (it keeps
// one chip of margin, :150) are skipped, which is only legitimate when they are silent -- the driver
// checks that and returns -3 otherwise.  Test streams therefore keep a silent gap in front of tags.
int ref_preamble_slicer_tt(const float *in, const float *inavg, uint64_t n, float rate,
                           float thr_db, uint64_t pad_items, float *bursts, uint64_t *tag_secs,
                           double *tag_frac, uint64_t *tag_item, uint64_t cap_tags,
                           uint64_t *n_tags, char *msgs, uint64_t msgs_cap, uint64_t *msgs_len,
                           uint64_t *n_msgs, uint64_t n_tt, const uint64_t *tt_offset,
                           const uint64_t *tt_secs, const double *tt_frac)
{
    gr::air_modes::preamble::sptr pre = gr::air_modes::preamble::make(rate, thr_db);
    for (uint64_t t = 0; t < n_tt; t++) {
        gr::tag_t g;
        g.offset = tt_offset[t];
        g.key = pmt::string_to_symbol("rx_time");
        g.value = pmt::make_tuple(pmt::from_uint64(tt_secs[t]), pmt::from_double(tt_frac[t]));
        pre->stub_in_tags.push_back(g);
    }
    const unsigned hist = pre->history();
    const uint64_t K = n + (hist - 1) + pad_items;
    std::vector<float> a(K + 64, 0.0f), b(K + 64, 0.0f);
    memcpy(a.data() + (hist - 1), in, n * sizeof(float));
    memcpy(b.data() + (hist - 1), inavg, n * sizeof(float));

    std::vector<float> stream;          // concatenated 240-sample bursts
    std::vector<gr::tag_t> tags;
    std::vector<uint64_t> items;        // r + i at which each tag was produced
    float out[240 * 4];
    uint64_t r = 0;
    for (;;) {
        if (K <= r) break;
        uint64_t win = K - r;
        bool cut = false;                   // window ends at the next rx_time tag
        for (uint64_t t = 0; t < n_tt; t++)
            if (tt_offset[t] > r && tt_offset[t] - r < win) { win = tt_offset[t] - r; cut = true; }
        gr_vector_int nin(2, (int)std::min<uint64_t>(win, 0x7fffffff));
        gr_vector_const_void_star ins(2);
        ins[0] = a.data() + r;
        ins[1] = b.data() + r;
        gr_vector_void_star outs(1, out);
        pre->stub_nitems_read = r;
        pre->stub_nitems_written = stream.size();
        pre->stub_consumed = 0;
        size_t ntag0 = pre->stub_out_tags.size();
        int produced = pre->general_work(240, nin, ins, outs);
        if (produced > 0) stream.insert(stream.end(), out, out + produced);
        for (size_t t = ntag0; t < pre->stub_out_tags.size(); t++) tags.push_back(pre->stub_out_tags[t]);
        if (produced == 0 && pre->stub_consumed == 0) {
            if (!cut) break;
            for (uint64_t k = r; k < r + win + 16; k++)      // skipping is only sound over silence
                if (a[k] != 0.0f || b[k] != 0.0f) return -3;
            r += win;
            continue;
        }
        r += (uint64_t)pre->stub_consumed;
    }

    *n_tags = tags.size();
    for (size_t t = 0; t < tags.size() && t < cap_tags; t++) {
        memcpy(bursts + t * 240, stream.data() + tags[t].offset, 240 * sizeof(float));
        tag_secs[t] = pmt::to_uint64(pmt::tuple_ref(tags[t].value, 0));
        tag_frac[t] = pmt::to_double(pmt::tuple_ref(tags[t].value, 1));
        tag_item[t] = tags[t].offset;
    }

    gr::msg_queue::sptr q = gr::msg_queue::make();
    gr::air_modes::slicer::sptr sl = gr::air_modes::slicer::make(q);
    size_t nb = stream.size();
    stream.resize(nb + 2048, 0.0f);      // room for the slicer's look-ahead margin
    sl->stub_in_tags = tags;
    sl->stub_nitems_read = 0;
    gr_vector_const_void_star sins(1, stream.data());
    gr_vector_void_star souts;
    sl->work((int)(nb + 960), sins, souts);

    uint64_t w = 0;
    for (const std::string &m : q->stub_msgs) {
        if (w + m.size() + 1 > msgs_cap) return -2;
        memcpy(msgs + w, m.data(), m.size());
        w += m.size();
        msgs[w++] = '\n';
    }
    *msgs_len = w;
    *n_msgs = q->stub_msgs.size();
    return 0;
}

// The reference's slicer alone (lib/slicer_impl.cc:102-198) over nb bursts of 240 floats handed in by the caller, burst k
// tagged "preamble_found" = (secs[k], frac[k]) at item 240 k.  msgs: '\n'-separated texts of the accepted ones, in order;
// accepted[k] = 1 where burst k produced a message (matched up by the order of the tags: the slicer visits them in order
// and posts at most one message per tag).  One work() call per `batch` bursts on a fresh window (the member ostringstream,
// and with it the precision quirk of the first message, lives as long as the block).  Returns 0 on success.
int ref_slicer(const float *bursts, uint64_t nb, const uint64_t *secs, const double *frac, uint8_t *accepted,
               char *msgs, uint64_t msgs_cap, uint64_t *msgs_len, uint64_t *n_msgs)
{
    gr::msg_queue::sptr q = gr::msg_queue::make();
    gr::air_modes::slicer::sptr sl = gr::air_modes::slicer::make(q);
    const uint64_t batch = 4096;
    uint64_t w = 0, total = 0;
    std::vector<float> stream;
    for (uint64_t b0 = 0; b0 < nb; b0 += batch) {
        const uint64_t cnt = std::min<uint64_t>(batch, nb - b0);
        stream.assign(bursts + b0 * 240, bursts + (b0 + cnt) * 240);
        stream.resize(cnt * 240 + 2048, 0.0f);      // room for the slicer's look-ahead margin
        for (uint64_t k = 0; k < cnt; k++) {
            // one tag at a time: which bursts were accepted is then known exactly
            sl->stub_in_tags.clear();
            gr::tag_t g;
            g.offset = k * 240;
            g.key = pmt::string_to_symbol("preamble_found");
            g.value = pmt::make_tuple(pmt::from_uint64(secs[b0 + k]), pmt::from_double(frac[b0 + k]));
            sl->stub_in_tags.push_back(g);
            sl->stub_nitems_read = 0;
            gr_vector_const_void_star sins(1, stream.data());
            gr_vector_void_star souts;
            const size_t before = q->stub_msgs.size();
            sl->work((int)(cnt * 240 + 960), sins, souts);
            const size_t after = q->stub_msgs.size();
            if (after > before + 1) return -4;
            accepted[b0 + k] = after > before ? 1 : 0;
        }
        for (const std::string &m : q->stub_msgs) {
            if (w + m.size() + 1 > msgs_cap) return -2;
            memcpy(msgs + w, m.data(), m.size());
            w += m.size();
            msgs[w++] = '\n';
        }
        total += q->stub_msgs.size();
        q->stub_msgs.clear();
    }
    *msgs_len = w;
    *n_msgs = total;
    return 0;
}

int ref_preamble_slicer(const float *in, const float *inavg, uint64_t n, float rate,
                        float thr_db, uint64_t pad_items, float *bursts, uint64_t *tag_secs,
                        double *tag_frac, uint64_t *tag_item, uint64_t cap_tags,
                        uint64_t *n_tags, char *msgs, uint64_t msgs_cap, uint64_t *msgs_len,
                        uint64_t *n_msgs)
{
    return ref_preamble_slicer_tt(in, inavg, n, rate, thr_db, pad_items, bursts, tag_secs, tag_frac, tag_item,
                                  cap_tags, n_tags, msgs, msgs_cap, msgs_len, n_msgs, 0, nullptr, nullptr, nullptr);
}

} // extern "C"
