// Stand-in for the slice of the GNU Radio 3.8 runtime API that
// /root/reference/lib/{preamble_impl,slicer_impl}.cc touch.  Written for this repo's
// oracle/_ref build only: a deterministic, single-threaded "scheduler" (ref_driver.cc)
// pokes the public stub_* members below instead of GR's buffers and tag plumbing.
// TEST INFRASTRUCTURE ONLY -- never part of the product.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
}

// ---------------------------------------------------------------- pmt ----
namespace pmt {
struct pmt_base {
    enum kind_t { K_NIL, K_SYM, K_U64, K_F64, K_TUPLE } kind = K_NIL;
    std::string sym;
    uint64_t u64 = 0;
    double f64 = 0.0;
    std::vector<std::shared_ptr<pmt_base>> items;
};
typedef std::shared_ptr<pmt_base> pmt_t;

inline pmt_t stub_make(pmt_base::kind_t k) { pmt_t p(new pmt_base); p->kind = k; return p; }
static const pmt_t PMT_NIL = stub_make(pmt_base::K_NIL);
inline pmt_t string_to_symbol(const std::string &s) { pmt_t p = stub_make(pmt_base::K_SYM); p->sym = s; return p; }
inline pmt_t intern(const std::string &s) { return string_to_symbol(s); }
inline bool is_symbol(const pmt_t &p) { return p && p->kind == pmt_base::K_SYM; }
inline std::string symbol_to_string(const pmt_t &p) { return p ? p->sym : std::string(); }
inline pmt_t from_uint64(uint64_t v) { pmt_t p = stub_make(pmt_base::K_U64); p->u64 = v; return p; }
inline uint64_t to_uint64(const pmt_t &p) { return p->u64; }
inline pmt_t from_double(double v) { pmt_t p = stub_make(pmt_base::K_F64); p->f64 = v; return p; }
inline double to_double(const pmt_t &p) { return p->f64; }
inline pmt_t make_tuple(const pmt_t &a, const pmt_t &b)
{
    pmt_t p = stub_make(pmt_base::K_TUPLE);
    p->items.push_back(a);
    p->items.push_back(b);
    return p;
}
inline pmt_t tuple_ref(const pmt_t &t, size_t i) { return t->items.at(i); }
} // namespace pmt

typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {

struct tag_t {
    uint64_t offset = 0;
    pmt::pmt_t key;
    pmt::pmt_t value;
    pmt::pmt_t srcid;
};

class io_signature
{
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int, int, int) { return sptr(new io_signature); }
    static sptr make2(int, int, int, int) { return sptr(new io_signature); }
};

class block
{
public:
    // --- driver-visible state (the stub "scheduler") ---
    uint64_t stub_nitems_read = 0;
    uint64_t stub_nitems_written = 0;
    int stub_consumed = 0;
    int stub_output_multiple = 1;
    unsigned stub_history = 1;
    std::vector<tag_t> stub_in_tags;   // tags visible on input 0
    std::vector<tag_t> stub_out_tags;  // tags the block added on output 0

    virtual ~block() {}
    virtual int general_work(int, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &) { return 0; }

    std::string name() const { return d_name; }
    long unique_id() const { return d_uid; }
    void set_output_multiple(int m) { stub_output_multiple = m; }
    void set_history(unsigned h) { stub_history = h; }
    unsigned history() const { return stub_history; }
    uint64_t nitems_read(unsigned) { return stub_nitems_read; }
    uint64_t nitems_written(unsigned) { return stub_nitems_written; }
    void consume_each(int n) { stub_consumed = n; }
    void add_item_tag(unsigned, uint64_t offset, const pmt::pmt_t &key, const pmt::pmt_t &value,
                      const pmt::pmt_t &srcid = pmt::PMT_NIL)
    {
        tag_t t;
        t.offset = offset; t.key = key; t.value = value; t.srcid = srcid;
        stub_out_tags.push_back(t);
    }
    void get_tags_in_range(std::vector<tag_t> &v, unsigned, uint64_t start, uint64_t end,
                           const pmt::pmt_t &key)
    {
        v.clear();
        for (const tag_t &t : stub_in_tags)
            if (t.offset >= start && t.offset < end && pmt::is_symbol(t.key) &&
                pmt::symbol_to_string(t.key) == pmt::symbol_to_string(key))
                v.push_back(t);
    }

protected:
    block() : d_name("stub"), d_uid(0) {}
    block(const std::string &name, io_signature::sptr, io_signature::sptr) : d_name(name), d_uid(next_uid()++) {}

private:
    static long &next_uid() { static long u = 0; return u; }
    std::string d_name;
    long d_uid;
};

class sync_block : public block
{
public:
    virtual int work(int, gr_vector_const_void_star &, gr_vector_void_star &) { return 0; }

protected:
    sync_block() {}
    sync_block(const std::string &name, io_signature::sptr a, io_signature::sptr b) : block(name, a, b) {}
};

class message
{
public:
    typedef std::shared_ptr<message> sptr;
    static sptr make_from_string(const std::string &s, long = 0, double = 0, double = 0)
    {
        sptr m(new message);
        m->d_text = s;
        return m;
    }
    std::string to_string() const { return d_text; }

private:
    std::string d_text;
};

class msg_queue
{
public:
    typedef std::shared_ptr<msg_queue> sptr;
    static sptr make(unsigned = 0) { return sptr(new msg_queue); }
    void handle(message::sptr m) { stub_msgs.push_back(m->to_string()); }
    void insert_tail(message::sptr m) { handle(m); }
    std::vector<std::string> stub_msgs;
};

} // namespace gr

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
} // namespace gnuradio
