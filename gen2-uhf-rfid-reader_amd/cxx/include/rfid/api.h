// rfid/api.h -- export macro of the gr::rfid blocks (the role of the reference's include/rfid/api.h:27-31)
#ifndef INCLUDED_RFID_API_H
#define INCLUDED_RFID_API_H
#define RFID_BLOCK_API __attribute__((visibility("default")))
#endif
